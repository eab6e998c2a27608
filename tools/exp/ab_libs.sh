#!/bin/bash
# A/B of builds of libvelocity_hip.so on ONE box (box-to-box spread is ~2 %, a warmed-up box repeats to ~0.5 %): alternates the libraries, 3 rounds.
# The product library velocity_amd/libvelocity_hip.so is NEVER written: variants live under _exp/ and are loaded through the explicit
# VH_LIB override (velocity_amd/_lib.py), which the bench line reports as build.override.
# Prepare here (CPU container), one per variant:   <edit csrc>;  python -m velocity_amd._build --out=_exp/lib_<name>.so
# ("head" = the product library itself) then:      gpurun -- bash tools/exp/ab_libs.sh head a b
# (the first run on a fresh box is the slow one: ignore round 1 of the first variant)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = head ]; then unset VH_LIB; else export VH_LIB=$PWD/_exp/lib_$v.so; fi
    python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['build']['build_id'], j['value'], j['roofline']['lk_us_per_launch'], [round(k['us_per_step']) for k in j['roofline_detail']['kernels']])"
  done
done
