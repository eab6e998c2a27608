S=256
R=/root/repo; cd /tmp; export TMPDIR=/tmp
for tag in head unipin; do
  if [ $tag = head ]; then unset VH_LIB; else export VH_LIB=$R/_exp/lib_unipin.so; fi
  OUT=$R/gpurun_out/pmc3_$tag; rm -rf $OUT; mkdir -p $OUT
  i=0
  for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_IFETCH_LEVEL"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --kernel-include-regex "k_lk3" --pmc $P --output-format csv -d $OUT/p$i -- python $R/bench.py --streams $S --steps 4 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --verify-frames 0 > $OUT/p$i.log 2>&1
  done
  python - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
json.dump({c: round(sum(v[len(v)//3:]) / len(v[len(v)//3:])) for c, v in acc.items()}, open("$R/gpurun_out/pmc3_$tag.json", "w"), indent=1)
PY
  find $OUT -name "*.csv" -delete
done
python - <<PY
import json
a=json.load(open("$R/gpurun_out/pmc3_head.json")); b=json.load(open("$R/gpurun_out/pmc3_unipin.json"))
for c in sorted(a): print(c, a[c], b.get(c), round(b.get(c,0)/max(a[c],1),3))
PY
