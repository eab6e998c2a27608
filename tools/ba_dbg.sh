for d in ${1:-0}; do
  echo -n "dbg=$d "; VH_BA_DBG=$d python bench.py --only-ba 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['ms_per_window_iter'] for k,v in j['by_windows'].items()})"
done
