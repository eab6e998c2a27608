#!/bin/bash
# Why the stream-interleaved block order helps the coarse LK launches: L2 (TCC) hit / miss / request counters of k_lk_o in natural order (VH_LKO_G=1) and in
# the default order.  Run on the GPU box; writes gpurun_out/lk_order_pmc.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/lk_order_pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for tag in natural default; do
  if [ $tag = natural ]; then export VH_LKO_G=1 VH_LK3_G=1; else unset VH_LKO_G VH_LK3_G; fi
  i=0
  for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --kernel-include-regex "k_lk_o|k_lk3" --pmc $P --output-format csv -d $OUT/$tag$i -- python $R/bench.py --streams 256 --steps 4 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --verify-frames 0 > $OUT/$tag$i.log 2>&1
  done
done
python - <<PY
import csv, glob, json
out = {}
for tag in ("natural", "default"):
    acc = {}
    for f in glob.glob("$OUT/%s*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out[tag] = {k: {c: round(sum(v[len(v) // 3:]) / len(v[len(v) // 3:])) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$R/gpurun_out/lk_order_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -delete
