#!/bin/bash
# A/B of two builds of libvelocity_hip.so on ONE box (box-to-box spread is ~2 %, a warmed-up box repeats to ~0.5 %): alternates the libraries, 3 rounds.
# Prepare here (CPU container):  build variant A, cp velocity_amd/libvelocity_hip.so _ab/lib_a.so; build variant B, cp ... _ab/lib_b.so
# then:  gpurun -- bash tools/exp/ab_libs.sh a b      (the first run on a fresh box is the slow one: ignore round 1 of the first variant)
cd /root/repo
cp velocity_amd/libvelocity_hip.so _ab/lib_restore.so
for r in 1 2 3; do
  for v in "$@"; do
    cp _ab/lib_$v.so velocity_amd/libvelocity_hip.so
    python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['value'], j['roofline']['lk_us_per_launch'], [round(k['us_per_step']) for k in j['roofline']['kernels']])"
  done
done
cp _ab/lib_restore.so velocity_amd/libvelocity_hip.so
