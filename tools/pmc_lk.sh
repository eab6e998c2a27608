#!/bin/bash
# SQ counters of the LK kernels only (two rocprofv3 --pmc passes); run on the GPU box.  Usage: bash tools/pmc_lk.sh [streams]
S=${1:-128}
R=/root/repo; OUT=$R/gpurun_out/pmc_lk; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "${VH_PMC_REGEX:-k_lk3|k_lk_o|k_lk_q}" --pmc $P --output-format csv -d $OUT/p$i -- python $R/bench.py --streams $S --groups 1 --steps 4 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    if "SQ_INSTS_VALU" in d and d.get("SQ_WAVES"):
        d["valu_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over the 1024 SIMDs; SQ_BUSY_CYCLES: summed over 32 shader engines
        d["valu_util"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / (8.0 * d["SQ_BUSY_CYCLES"]), 4) if d.get("SQ_BUSY_CYCLES") else None
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -delete
