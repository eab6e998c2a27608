# stream-interleave group size sweeps of the batched LK kernels (experiment: the VH_*_G variables are read by an experimental build loaded through VH_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export VH_LIB=$PWD/_exp/lib_rt.so
run() { python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 --streams ${S:-256} 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['roofline']['lk_us_per_launch'])"; }
for r in 1 2; do
  for g in 1 8 16 24 32 64 128; do VH_LK3_G=$g run "lk3 G=$g"; done
  for g in 32 40 48 64 96 128 256; do VH_LKO_G=$g run "lko G=$g"; done
done
