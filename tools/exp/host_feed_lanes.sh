#!/bin/bash
# PCIe-inclusive frame rate (bench.py --host-frames, 64 streams of 1080p) vs the number of HIP streams one HostFrameFeeder upload is split over; run on the GPU box
for l in 1 2 4 8; do echo -n "lanes $l: "; VH_FEEDER_LANES=$l python bench.py --streams 64 --steps 60 --warmup 10 --cpu-seconds 0 --no-ba --no-extras --min-seconds 1 --host-frames --verify-frames 0 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; done
