#!/bin/bash
# Calibrates the VALU-instruction model of the fine-stage LK kernel: rocprofv3 --pmc SQ_INSTS_VALU of k_lk3<51,1,4> at three Newton-iteration caps,
# a least-squares fit of   wave instructions per launch = A x template set-ups + B x Newton iterations   against the kernel's own counters (the bench
# line prints them), and a check run (the roll scene) that states the model's error.  Run on the GPU box; writes gpurun_out/lk_valu_model.json
# (copy to profiles/r03_lk_valu_model.json).  Usage: bash tools/pmc_lk_calib.sh [streams]
S=${1:-256}
R=/root/repo; OUT=$R/gpurun_out/pmc_calib; rm -rf $OUT $R/gpurun_out/lk_valu_model.json; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {  # tag, extra bench flags
  rocprofv3 --kernel-trace --kernel-include-regex "k_lk3|k_lk_o|k_lk_q" --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/$1 -- \
    python $R/bench.py --streams $S --groups 1 --steps 4 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail $OUT/$1.json --verify-frames 0 $2 > $OUT/$1.log 2>&1
}
run default ""
run cap1 "--fine-max-count 1"
run cap3 "--fine-max-count 3"
run roll "--scene roll"
run ccap1 "--coarse-max-count 1"
run ccap2 "--coarse-max-count 2"
python - <<PY
import csv, glob, json
import numpy as np
pts, cpts = {}, {}
for tag in ("default", "cap1", "cap3", "roll", "ccap1", "ccap2"):
    j = json.load(open("$OUT/%s.json" % tag))  # the FULL record of that run (bench.py --detail); stdout carries only the compact line
    rf = j["roofline_detail"]
    vals, cvals = {}, {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_lk3" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            elif r["Counter_Name"] == "SQ_INSTS_VALU":  # the two coarse launches of a step alternate: stage 1 (quarter scale), stage 2 (ROI), in dispatch order
                cvals.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    v = vals["SQ_INSTS_VALU"]
    v = v[len(v) // 3:]  # steady state (the first launches follow the frame-0 state)
    if tag in ("default", "cap1", "cap3", "roll"):
        pts[tag] = dict(kernel=rf["kernel"].split(" (")[0], setups=rf["setups_per_launch"], iters=rf["newton_iters_per_launch"], wave_instr=float(np.mean(v)),
                        launches=len(v), waves=float(np.mean(vals["SQ_WAVES"])))
    if tag in ("default", "ccap1", "ccap2", "roll"):
        for kname, lst in cvals.items():
            lst = [c for _, c in sorted(lst)]
            lst = lst[2 * (len(lst) // 6):]  # steady state, starting on a stage-1 launch
            for stg in (0, 1):
                row = rf["kernels"][stg]
                cpts["%s_stage%d" % (tag, stg + 1)] = dict(kernel=row["kernel"].split(" (")[0], setups=row["setups_per_launch"], iters=row["newton_iters_per_launch"],
                                                          wave_instr=float(np.mean(lst[stg::2])), launches=len(lst[stg::2]))
fit = ["default", "cap1", "cap3"]
A = np.array([[pts[t]["setups"], pts[t]["iters"]] for t in fit], float)
b = np.array([pts[t]["wave_instr"] for t in fit])
(a_, b_), *_ = np.linalg.lstsq(A, b, rcond=None)
res = {t: dict(pts[t], model=float(a_ * pts[t]["setups"] + b_ * pts[t]["iters"])) for t in pts}
for t in res: res[t]["rel_err"] = res[t]["model"] / res[t]["wave_instr"] - 1.0
tol = max(0.02, 1.5 * max(abs(res[t]["rel_err"]) for t in res))
out = dict(kernel=pts["default"]["kernel"], streams=$S, wave_instr_per_setup=float(a_), wave_instr_per_newton_iter=float(b_), tolerance=round(float(tol), 4),
           _comment="wave instructions (SQ_INSTS_VALU) per launch = A x set-ups + B x Newton iterations; fitted on default / cap1 / cap3, checked on the roll scene; "
                    "counters per launch from the bench line of the same run (in-kernel), PMC values = mean over the steady-state launches",
           points=res)
# the coarse kernel (both launches of a step, three iteration caps): same model, fitted on default / ccap1 / ccap2, checked on the roll scene
if cpts:
    fitc = [k for k in cpts if not k.startswith("roll")]
    Ac = np.array([[cpts[t]["setups"], cpts[t]["iters"]] for t in fitc], float)
    bc = np.array([cpts[t]["wave_instr"] for t in fitc])
    (ac_, bc_), *_ = np.linalg.lstsq(Ac, bc, rcond=None)
    resc = {t: dict(cpts[t], model=float(ac_ * cpts[t]["setups"] + bc_ * cpts[t]["iters"])) for t in cpts}
    for t in resc: resc[t]["rel_err"] = resc[t]["model"] / resc[t]["wave_instr"] - 1.0
    out["coarse"] = dict(kernel=cpts[fitc[0]]["kernel"], wave_instr_per_setup=float(ac_), wave_instr_per_newton_iter=float(bc_),
                         tolerance=round(float(max(0.02, 1.5 * max(abs(resc[t]["rel_err"]) for t in resc))), 4),
                         _comment="wave instructions per launch = A x set-ups + B x Newton iterations with PER-TRACK counters (8 / 4 tracks share a wavefront: A and B are per track, "
                                  "and B carries the idle lanes of wavefronts whose tracks need different iteration counts)", points=resc)
json.dump(out, open("$R/gpurun_out/lk_valu_model.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -delete
