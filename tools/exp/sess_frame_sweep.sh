for s in 2 8 32 64 128 256; do python bench.py --streams $s --steps 40 --warmup 5 --no-ba --no-extras --cpu-seconds 0 --min-seconds 0 --verify-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print($s, j['value'], [(k['kernel'][:12], k['us_per_step']) for k in j['roofline_detail']['kernels'] if 'sess' in k['kernel'] or 'ransac' in k['kernel']])"; done
