#!/bin/bash
# A/B of the workgroup shape of k_roi_warp (VH_RW_BLK) and of the 8-rows-per-thread k_pyr_down (VH_PD_BLK) on ONE box: 0 = 64 x 4 threads, 1 = 16 x 16,
# 2 = 8 x 32, 3 = 16 x 8, 4 = 32 x 8.  Parity tests per shape, then alternating rounds of the default bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 1 2 3 4; do
  VH_RW_BLK=$v VH_PD_BLK=$v python -m pytest tests/test_gpu_klt.py -x -q -k "pyr_down or remap or klt_regional or baseline_sizes or klt_main_bit_exact_all" 2>&1 | tail -1
done
run() { VH_RW_BLK=$1 VH_PD_BLK=$2 python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('rw$1 pd$2', j['value'], [(k['kernel'][:10], round(k['us_per_step'])) for k in j['roofline_detail']['kernels'][2:4]])"; }
for r in 1 2; do
  run 0 0; run 1 0; run 2 0; run 3 0; run 4 0; run 1 1; run 1 4
done 2>&1 | tee gpurun_out/img_blk_sweep.log
