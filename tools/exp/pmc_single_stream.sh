#!/bin/bash
# SQ counters of the one-stream LK launches (k_lk_strip<15>, k_lk3<51,2,4>): where does a 60 us launch of 2000 one-wavefront workgroups spend its time?
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_s1; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_IFETCH"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "k_lk" --pmc $P --output-format csv -d $OUT/p$i -- python $R/bench.py --streams 1 --steps 30 --warmup 10 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -delete; tail -3 $OUT/p2.log
