# small batches: natural order (G=1) against the interleaved + pinned order
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export VH_LIB=$PWD/_exp/lib_rt.so
run() { python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 --streams $S --min-seconds 1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('S=$S $1', j['value'], j['roofline']['lk_us_per_launch'], j['roofline_detail']['kernels'][0]['kernel'][:8])"; }
for S in 4 8 16 32 64 100; do
  VH_LK3_G=1 VH_LKO_G=1 VH_LKQ_G=1 run "all natural"
  VH_LK3_G=8 VH_LKO_G=1 VH_LKQ_G=1 run "lk3 G=8"
  VH_LK3_G=8 VH_LKO_G=64 VH_LKQ_G=64 run "lk3 G=8 coarse G=64"
done
