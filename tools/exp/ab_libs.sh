#!/bin/bash
# A/B of builds of libvelocity_hip.so on ONE box (box-to-box spread is ~2 %, a warmed-up box repeats to ~0.5 %): alternates the libraries, 3 rounds.
# The product library velocity_amd/libvelocity_hip.so is NEVER written: variants live under _exp/ and are loaded through the explicit
# VH_LIB override (velocity_amd/_lib.py), which the bench line reports as build.override.
# Prepare here (CPU container), one per variant:   <edit csrc>;  python -m velocity_amd._build --out=_exp/lib_<name>.so
# ("head" = the product library itself) then:      gpurun -- bash tools/exp/ab_libs.sh head a b
# (the first run on a fresh box is the slow one: ignore round 1 of the first variant)
# AB_ARGS: extra bench arguments (e.g. "--streams 4"); AB_ROUNDS: rounds (default 3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    if [ "$v" = head ]; then unset VH_LIB; else export VH_LIB=$PWD/_exp/lib_$v.so; fi
    python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 --detail /dev/null $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['build_id'], j['value'], j['ms_per_step'], j['roofline']['lk_kernels'], j['roofline']['lk_us_per_launch'])"
  done
done
