#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, stream sweep, rocprofv3 kernel stats and the two HBM-traffic PMC passes.
# Usage: bash tools/collect_profiles.sh <streams> ; results under gpurun_out/prof/
S=${1:-128}
R=/root/repo
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --streams $S > $OUT/bench_default.json 2> $OUT/bench_default.err
for s in 1 2 4 8 16 32 64 128 256; do
  python $R/bench.py --streams $s --steps 60 --warmup 10 --cpu-seconds 0 --no-ba 2>/dev/null | tail -1 > $OUT/sweep_$s.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --streams $S --steps 40 --warmup 5 --cpu-seconds 0 --no-ba > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete   # tens of MB; the per-kernel summary is what is kept
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --kernel-include-regex 'k_lk3|k_lk_q|k_lk_strip|k_pyr_down|k_roi_warp' --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --streams $S --steps 6 --warmup 2 --cpu-seconds 0 --no-ba > $OUT/pmc_$c.log 2>&1
  find $OUT/pmc_$c -name "*kernel_trace.csv" -delete
done
# BA (C5): per-kernel stats of bench.bench_ba()
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba -- python -c "
import sys; sys.path.insert(0, '$R')
import bench, json
print(json.dumps(bench.bench_ba()))" > $OUT/ba.log 2>&1
find $OUT/ba -name "*kernel_trace.csv" -delete
# PCIe-inclusive rate (frames uploaded from pinned host memory every step)
for s in 1 8 64; do
  python $R/bench.py --streams $s --steps 60 --warmup 10 --cpu-seconds 0 --no-ba --host-frames 2>/dev/null | tail -1 > $OUT/hostframes_$s.json
done
du -sh $OUT; tail -2 $OUT/bench_default.err; ls -R $OUT | head -40
