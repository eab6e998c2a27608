#!/bin/bash
# per-kernel time of the default frame step (rocprofv3 --kernel-trace --stats); run on the GPU box.  Usage: bash tools/kernel_stats.sh [streams]
S=${1:-128}
R=/root/repo; OUT=$R/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --streams $S --steps 40 --warmup 5 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/log.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(("k_", "void k_")))
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct_of_k {100*float(r["TotalDurationNs"])/tot:5.1f}')
PY
