#!/bin/bash
# launch timeline of ONE steady-state frame step (rocprofv3 --kernel-trace); run on the GPU box.  Usage: bash tools/step_timeline.sh [streams]
S=${1:-128}
R=/root/repo; OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --streams $S --steps 30 --warmup 5 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if k[2].startswith("k_klt_setup")]
a, b = idx[-10], idx[-9]
t0 = ks[a][0]
out = dict(step_us=round((ks[b][0] - t0) / 1e3, 1),
           launches=[dict(t_us=round((s - t0) / 1e3, 1), dur_us=round((e - s) / 1e3, 1), kernel=n.split("(")[0].replace("void ", "")) for s, e, n in ks[a:b]])
json.dump(out, open("$OUT/step.json", "w"), indent=1)
print("step_us", out["step_us"])
for l in out["launches"]: print(f'{l["t_us"]:9.1f} {l["dur_us"]:9.1f}  {l["kernel"]}')
PY
find $OUT -name "*.csv" -delete
