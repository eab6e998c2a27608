"""rpy2dcm / dcm2rpy (utils/transforms.py:7-23,51-57): 3x3 host scalar math (the per-point work runs on the GPU)."""
import math

import numpy as np


def rpy2dcm(rpy):
    """[roll, pitch, yaw] -> direction cosine matrix, used as a RIGHT multiplier (transforms.py:7-23)."""
    sr, cr = math.sin(rpy[0]), math.cos(rpy[0])
    sp, cp = math.sin(rpy[1]), math.cos(rpy[1])
    sy, cy = math.sin(rpy[2]), math.cos(rpy[2])
    return np.array([[cp * cy, sr * sp * cy - cr * sy, cr * sp * cy + sr * sy],
                     [cp * sy, sr * sp * sy + cr * cy, cr * sp * sy - sr * cy],
                     [-sp, sr * cp, cr * cp]])


def dcm2rpy(R):
    """Direction cosine matrix -> [roll, pitch, yaw] (transforms.py:51-57)."""
    return np.array([math.atan(R[2, 1] / R[2, 2]), math.asin(-R[2, 0]), math.atan2(R[1, 0], R[0, 0])])
