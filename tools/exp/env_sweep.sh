#!/bin/bash
# A/B of ONE library under different values of an experiment environment variable on ONE box.
# usage: env_sweep.sh VAR v1 v2 ...   (three alternating rounds of the default bench without extras)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v', j['value'], j['roofline']['lk_us_per_launch'], [round(k['us_per_step']) for k in j['roofline_detail']['kernels']])"
  done
done
