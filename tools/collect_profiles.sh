#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, stream sweep, rocprofv3 kernel stats, the HBM-traffic PMC passes, the SQ passes of the LK and BA
# kernels, the single-stream launch timeline, the PCIe-inclusive rates and the VALU issue-rate micro-benchmark.
# Usage: bash tools/collect_profiles.sh <streams> ; results under gpurun_out/prof/ ; then: python tools/summarize_profiles.py r02
S=${1:-256}
R=/root/repo
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# gate: nothing is profiled unless the loaded binary IS the tree's source and the whole GPU suite is green on it
cd $R
python - > $OUT/gate.json <<PY
import json, sys
sys.path.insert(0, "$R")
from velocity_amd import _build, _lib
i = _lib.build_info()
i["tree_build_id"] = _build.build_id()
i["ok"] = bool(i["matches_source"] and not i["override"] and i["build_id"] == i["tree_build_id"])
print(json.dumps(i))
PY
if ! grep -q '"ok": true' $OUT/gate.json; then echo "collect_profiles: the library is not built from this tree:"; cat $OUT/gate.json; exit 3; fi
if [ -z "$VH_SKIP_SUITE" ]; then
  python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  rc=$?
  tail -3 $OUT/pytest_gpu.log
  if [ $rc -ne 0 ]; then echo "collect_profiles: pytest -m gpu is RED (rc $rc): no profiles are written"; exit 4; fi
fi
cd /tmp && export TMPDIR=/tmp
# first: re-fit the instruction-cost model of the fine LK kernel for THIS build (the bench line prices its live counters with it); the file travels back
# as gpurun_out/prof/lk_valu_model.json and summarize_profiles.py files it as profiles/rNN_lk_valu_model.json
bash $R/tools/pmc_lk_calib.sh $S > $OUT/lk_calib.log 2>&1
M=$(ls $R/profiles/r[0-9][0-9]_lk_valu_model.json 2>/dev/null | tail -1)
if ! [ -s $R/gpurun_out/lk_valu_model.json ]; then echo "collect_profiles: the instruction-cost fit FAILED (see lk_calib.log): the bench line will price with the previous round's model"; tail -5 $OUT/lk_calib.log; fi
if [ -s $R/gpurun_out/lk_valu_model.json ]; then cp $R/gpurun_out/lk_valu_model.json $OUT/lk_valu_model.json; [ -n "$M" ] && cp $R/gpurun_out/lk_valu_model.json $M; fi
python $R/bench.py --streams $S --verbose --detail $OUT/bench_default.json > $OUT/bench_default.line 2> $OUT/bench_default.err
for s in 1 2 4 8 16 32 64 128 256; do
  python $R/bench.py --streams $s --steps 60 --warmup 10 --cpu-seconds 0 --no-ba --no-extras --min-seconds 1 --detail /dev/null 2>/dev/null | tail -1 > $OUT/sweep_$s.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --streams $S --steps 40 --warmup 5 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete   # tens of MB; the per-kernel summary is what is kept
# (counter passes run ONE session -- --groups 1 -- so that a launch is 256 streams and launches do not overlap; the stats pass above and the bench line run the
# headline's own shape, two sessions on two HIP streams: their per-launch durations agree with each other)
# HBM traffic of EVERY library kernel of a step (roofline.step_hbm sums them): FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --kernel-include-regex '^(void )?k_' --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --streams $S --groups 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/pmc_$c.log 2>&1
  find $OUT/pmc_$c -name "*kernel_trace.csv" -delete
done
# the loads that look like the reference's data: both episode legs + the kernel stats of the hard scene
python $R/bench.py --only-leg hard_scene:$S > $OUT/hard_scene.json 2> $OUT/hard_scene.err
python $R/bench.py --only-leg hard_scene:8 > $OUT/hard_scene_8.json 2>> $OUT/hard_scene.err
python $R/bench.py --only-leg real_texture:$S > $OUT/real_texture.json 2>> $OUT/hard_scene.err
python $R/bench.py --only-leg real_texture:8 > $OUT/real_texture_8.json 2>> $OUT/hard_scene.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hard_stats -- python $R/bench.py --only-leg hard_scene:$S --verify-frames 0 > $OUT/hard_stats.log 2>&1
find $OUT/hard_stats -name "*kernel_trace.csv" -delete
# SQ counters of the LK kernels (two passes of <= 8 counters)
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex 'k_lk3|k_lk_o|k_lk_q' --pmc $P --output-format csv -d $OUT/sq_lk$i -- python $R/bench.py --streams $S --groups 1 --steps 4 --warmup 2 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/sq_lk$i.log 2>&1
  find $OUT/sq_lk$i -name "*kernel_trace.csv" -delete
done
# BA (C5): per-kernel stats of bench.bench_ba() (1, 8 and 64 windows) + SQ / MFMA counters of its kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba -- python $R/bench.py --only-ba > $OUT/ba.log 2>&1
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/ba/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_ba", "void k_ba"))]
# per window count (grid z... the window index is grid.y): average duration per kernel
acc = {}
for r in rows:
    nw = int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc.setdefault(nw, {}).setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
json.dump({str(nw): {k: dict(launches=len(v), avg_us=round(sum(v) / len(v) / 1e3, 2)) for k, v in d.items()} for nw, d in sorted(acc.items())},
          open("$OUT/ba_by_windows.json", "w"), indent=1)
PY
find $OUT/ba -name "*kernel_trace.csv" -delete
PB1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
PB2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$PB1" "$PB2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex 'k_ba_' --pmc $P --output-format csv -d $OUT/sq_ba$i -- python $R/bench.py --only-ba > $OUT/sq_ba$i.log 2>&1
  python - <<PY
import csv, glob, json
f = glob.glob("$OUT/sq_ba$i/**/*kernel_trace.csv", recursive=True)
dur = {}
if f:
    for r in csv.DictReader(open(f[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1))
json.dump(dur, open("$OUT/sq_ba${i}_dispatch.json", "w"))
PY
  find $OUT/sq_ba$i -name "*kernel_trace.csv" -delete
done
# BA by free cameras (one window, 5000 / 1000 points): per-stage times of an LM iteration from 19 to 255 cameras + kernel stats of the 128-camera window
BA_NF=20,26,43,51,65,97,129,201,256 python $R/tools/exp/ba_by_cameras.py 5000 1000 > $OUT/ba_by_cameras.jsonl 2> $OUT/ba_by_cameras.log
BA_NF=129 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba128 -- python $R/tools/exp/ba_by_cameras.py 5000 > $OUT/ba128.log 2>&1
find $OUT/ba128 -name "*kernel_trace.csv" -delete
# single stream: launch timeline of one steady-state frame
rocprofv3 --kernel-trace --output-format csv -d $OUT/s1 -- python $R/bench.py --streams 1 --steps 200 --warmup 20 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0 --detail /dev/null > $OUT/s1.log 2>&1
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if k[2].startswith("k_klt_setup")]
a, b = idx[150], idx[151]
t0 = ks[a][0]
json.dump(dict(_comment="rocprofv3 --kernel-trace of python bench.py --streams 1: the launches of ONE steady-state frame step (start offset, duration; "
                        "the tracer adds a few microseconds to every launch: bench.py reports the untraced step time)",
               step_us=round((ks[b][0] - t0) / 1e3, 1),
               launches=[dict(t_us=round((s - t0) / 1e3, 1), dur_us=round((e - s) / 1e3, 1), kernel=n.split("(")[0].replace("void ", "")) for s, e, n in ks[a:b]]),
          open("$OUT/s1_timeline.json", "w"), indent=1)
PY
rm -rf $OUT/s1
# PCIe-inclusive rate (frames uploaded from pinned host memory every step)
for s in 1 8 64; do
  python $R/bench.py --streams $s --steps 60 --warmup 10 --cpu-seconds 0 --no-ba --no-extras --min-seconds 1 --host-frames --detail /dev/null 2>/dev/null | tail -1 > $OUT/hostframes_$s.json
done
# VALU issue rates per instruction class
timeout 600 $R/tools/ubench/valu_rate > $OUT/valu_rate.json 2> $OUT/valu_rate.err
# f64 MFMA next to LDS operand reads (the step structure of k_ba_syrk_mfma / k_ba_chol_left)
[ -x $R/tools/ubench/mfma_lds ] && timeout 120 $R/tools/ubench/mfma_lds > $OUT/mfma_lds.json 2> $OUT/mfma_lds.err
du -sh $OUT; tail -2 $OUT/bench_default.err; ls $OUT | head -60
