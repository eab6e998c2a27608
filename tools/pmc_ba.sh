#!/bin/bash
# SQ counters of the BA kernels (passes of <= 8 counters), run on the GPU box: bash tools/pmc_ba.sh
R=/root/repo; OUT=$R/gpurun_out/pmc_ba; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex 'k_ba_' --pmc $P --output-format csv -d $OUT/p$i -- python $R/bench.py --only-ba > $OUT/p$i.log 2>&1
  find $OUT/p$i -name "*kernel_trace.csv" -delete
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    f = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not f: print("pass", i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        # the last launches are the 64-window batch (largest values): report the maximum per counter and the mean of the top 10 %
        print(k, {c: round(max(v)) for c, v in d.items()})
PY
