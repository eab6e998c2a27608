# small batches: natural order, pinned interleave, rotated (unpinned) interleave  (VH_*_G + 65536 = rotated; experimental build through VH_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export VH_LIB=$PWD/_exp/lib_rot.so
run() { python bench.py --no-ba --no-extras --cpu-seconds 0 --verify-frames 0 --streams $S --min-seconds 1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('S=$S $1', j['value'], j['roofline']['lk_us_per_launch'])"; }
for S in 4 8 16 64 256; do
  VH_LK3_G=1 VH_LKO_G=1 VH_LKQ_G=1 run "natural"
  VH_LK3_G=16 VH_LKO_G=64 VH_LKQ_G=64 run "pinned"
  VH_LK3_G=65552 VH_LKO_G=65600 VH_LKQ_G=65600 run "rotated"
done
