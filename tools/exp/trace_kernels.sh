#!/bin/bash
# usage: trace_kernels.sh <tag> <python script + args...>: per-launch kernel trace copied to gpurun_out/<tag>_kernel_trace.csv (GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python "$@" > /tmp/o_$tag.txt 2>&1
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" < /dev/null | head -1)
if [ -n "$t" ]; then cp $t /root/repo/gpurun_out/${tag}_kernel_trace.csv; else tail -5 /tmp/o_$tag.txt; fi
exit 0
