#!/bin/bash
# Occupancy sensitivity of the fine-stage LK kernel k_lk3<51,1,4> (168 VGPRs -> 3 wavefronts per SIMD, 7.7 KB of LDS per one-wave workgroup): extra
# dynamic LDS per workgroup (VH_LK_LDS_PAD) caps the resident workgroups per CU at 12 / 8 / 4 = 3 / 2 / 1 wavefronts per SIMD.  Run on the GPU box.
R=/root/repo
for pad in 0 11800 30000; do
  echo -n "VH_LK_LDS_PAD=$pad: "
  VH_LK_LDS_PAD=$pad python $R/bench.py --streams ${1:-256} --steps 20 --warmup 5 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('fine LK us/launch', r['lk_us_per_launch'][2], ' frames/s', j['value'], ' valu frac', r['frac'])"
done
