#!/bin/bash
# The checker's checker: the C oracle (oracle/klt_oracle.c) rebuilt with AddressSanitizer and run over (a) its own CPU tests and (b) the edge inputs the GPU
# tests feed it (fuzz, images smaller than the window, 1-3 tracks, blank frames, points outside the frame: tools/exp/oracle_asan_inputs.py).  Round 4 found a
# heap overflow this way (ko_bounding_rect on an empty track list) after the real-stills test segfaulted on the GPU box.  CPU only; ~1 min.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
gcc -O1 -g -mavx2 -fopenmp -ffp-contract=off -fPIC -shared -fsanitize=address -fno-omit-frame-pointer $R/oracle/klt_oracle.c -o /tmp/libklt_asan.so -lm
export KO_LIB=/tmp/libklt_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
cd $R
timeout 1500 python -m pytest tests/test_oracle_klt.py tests/test_oracle_stills.py tests/test_oracle_klt_sensitivity.py -x -q
timeout 1500 python tools/exp/oracle_asan_inputs.py
