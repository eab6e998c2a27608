#!/bin/bash
# ablation of k_ba_schur_mfma (VH_BA_DBG bits: 1 no MFMA, 2 no diag, 32 no Z, 4 no fetch, 16 no park, 8 no barrier): kernel time at 64 windows
R=/root/repo; cd /tmp; export TMPDIR=/tmp
for d in ${1:-0}; do
  OUT=$R/gpurun_out/prof_dbg; rm -rf $OUT; mkdir -p $OUT
  VH_BA_DBG=$d rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --only-ba > $OUT/ba.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "k_ba_schur" not in r["Kernel_Name"]: continue
    nw = int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)
    acc.setdefault(nw, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("dbg=$d schur us:", {nw: round(sum(v) / len(v) / 1e3, 1) for nw, v in sorted(acc.items())})
PY
  rm -rf $OUT
done
