"""CPU oracle of the reference's frame loop (vidExample.py:75-171, hot-path lines only).  TEST INFRASTRUCTURE ONLY.

Follows the driver statement by statement with the reference's dtypes (float32 B/S/P records, bool masks), using
oracle/klt_oracle.py for KLTmain and oracle/nls_oracle.py for the pose / MSV solves.  The two broken driver lines
(`im0` never assigned, vidExample.py:131-147) follow SURVEY Appendix B: im0 = previous frame.
"""
import numpy as np

from . import klt_oracle as KO
from . import nls_oracle as NO


class SessionOracle:
    def __init__(self, K32, frame0, p, p3, vp, t0, time0=0.0, frame_no=0.0, res0=0.0, nhist=20, lk_coarse=None, lk_fine=None,
                 msv_frame=5, native=False):
        n = nhist
        self.K = np.asarray(K32, np.float32)
        self.lkc, self.lkf = lk_coarse, lk_fine
        self.lib = KO.lib(native=native)
        self.p = np.asarray(p, np.float32).copy()
        self.p3 = np.asarray(p3, np.float64).copy()
        N0 = self.p.shape[0]
        self.B = np.zeros([n, 14], np.float32)  # vidExample.py:44
        self.S = np.zeros([n, 9], np.float32)  # :45
        self.B[0, 0:3] = t0  # :121
        self.B[0, 12], self.B[0, 13] = time0, frame_no
        self.vg = np.ones(N0, bool)  # :125
        self.vp = np.asarray(vp, bool).copy()  # :126
        self.P = np.full([5, N0, n], np.nan, np.float32)  # :128-129
        self.R = np.eye(3)  # :120
        self.im0, self.im0_small = np.asarray(frame0), None  # :131 (+ App. B)
        self.r, self.t0_time, self.i = np.float32(0), self.B[0, 12], 0
        self.msv_frame = msv_frame
        self.t = np.asarray(t0, np.float32)
        p_ = self.p[self.vp]  # :127
        self.P[0:2, self.vg, 0] = self.p.T  # :151-153
        self.P[2:4, self.vp, 0] = p_.T
        self.P[4, self.vg, 0] = 0
        self.S[0, :] = (0, 0, self.vg.sum(), res0, np.nan, 0, 0, 0, np.nan)
        self.residuals = res0

    def step(self, im, time_s, frame_no):
        self.i += 1
        i, B = self.i, self.B
        B[i, 12], B[i, 13] = time_s, frame_no
        p, v, self.im0_small = KO.klt_main(im, self.im0, self.im0_small, self.p, lk_coarse=self.lkc, lk_fine=self.lkf, L=self.lib)  # :134
        self.vg[self.vg] = v  # :135
        self.vp = self.vp & self.vg  # :136
        self.p = p
        t, R, residuals, p_ = NO.estimate_world_camera_pose(self.K, p[self.vp[self.vg]], self.p3[self.vp], R=self.R, findR=False)  # :139
        dt = B[i, 12] - B[i - 1, 12]  # :142
        dr = NO.l2(t + B[0, 0:3] - B[i - 1, 0:3])  # :143
        self.r = self.r + dr
        B[i, 3:6] = t
        B[i, 0:3] = B[0, 0:3] + t
        self.im0 = np.asarray(im)  # App. B intent of :147
        self.P[0:2, self.vg, i] = p.T  # :151-153
        self.P[2:4, self.vp, i] = p_.T
        self.P[4, self.vg, i] = i
        self.t, self.residuals = t, residuals
        if i == self.msv_frame:  # :155-160
            _x, p3hat = NO.msv1_t(self.K, self.P, B, self.vg, i)
            self.p3[self.vg] = p3hat - t
            self.vp = self.vg.copy()
        self.S[i, :] = (i, 0, self.vg.sum(), residuals, dt, B[i, 12] - self.t0_time, dr, self.r, dr / dt * np.float32(3.6))  # :164
        return t, residuals
