#!/bin/bash
# k_roi_warp in natural and in XCD-band tile order (VH_RW_XCD): FETCH_SIZE / WRITE_SIZE per launch (separate passes) and the launch time.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/rw_xcd; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    VH_RW_XCD=$v rocprofv3 --kernel-trace --kernel-include-regex 'k_roi_warp' --pmc $c --output-format csv -d $OUT/v${v}_$c -- python $R/bench.py --streams 256 --groups 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --verify-frames 0 --detail /dev/null > $OUT/v${v}_$c.log 2>&1
  done
done
python - <<PY
import csv, glob, json
out = {}
for v in (0, 1):
    row = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/v%d_%s/**/*counter_collection.csv" % (v, c), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "k_roi_warp" in r["Kernel_Name"]]
        vals = vals[len(vals) // 4:]
        row[c.lower() + "_kib_per_launch"] = round(sum(vals) / len(vals), 1)
        t = glob.glob("$OUT/v%d_%s/**/*kernel_trace.csv" % (v, c), recursive=True)[0]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "k_roi_warp" in r["Kernel_Name"]]
        row["us_per_launch_under_pmc"] = round(sum(d[len(d) // 4:]) / len(d[len(d) // 4:]) / 1e3, 1)
    row["bytes_per_launch"] = int((2 * row["fetch_size_kib_per_launch"] + row["write_size_kib_per_launch"]) * 1024)
    out["xcd_bands" if v else "natural"] = row
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/rw_xcd.json", "w"), indent=1)
PY
