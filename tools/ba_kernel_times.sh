#!/bin/bash
# per-kernel BA times by window count (rocprofv3 kernel trace of bench.py --only-ba); run on the GPU box
R=/root/repo; OUT=$R/gpurun_out/prof_ba; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --only-ba > $OUT/ba.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "k_ba" not in r["Kernel_Name"]: continue
    nw = int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc.setdefault(nw, {}).setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for nw, d in sorted(acc.items()):
    print(nw, {k: round(sum(v) / len(v) / 1e3, 1) for k, v in d.items()})
PY
rm -rf $OUT
