#!/bin/bash
# A/B of an environment switch of the library on ONE box: env_ab.sh VAR v1 v2 ... ; AB_ARGS = extra bench arguments; prints value + per-kernel us per step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; shift
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    env $VAR=$v python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 --groups 1 --detail /tmp/ab_detail.json $AB_ARGS > /dev/null 2>&1
    python -c "
import json; j=json.load(open('/tmp/ab_detail.json')); print('$VAR=$v', j['value'], j['ms_per_step'], {k['kernel'].split(' (')[0]: k['us_per_step'] for k in j['roofline_detail']['kernels']})"
  done
done
