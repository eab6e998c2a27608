#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of tools/ubench/fetch_calib (1 GiB each): the counters' calibration by access width.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- $R/tools/ubench/fetch_calib > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        continue
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            per.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        out.setdefault(k, {})[c + "_kib_per_GiB"] = [round(x, 1) for x in v]
        out[k][c + "_reported_over_actual"] = round(v[-1] / 2 ** 20, 4)
json.dump(out, open("$OUT/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
