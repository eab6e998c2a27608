#!/bin/bash
# kernel stats of tools/exp/ba_by_cameras.py under rocprofv3 (GPU box).  usage: ba_kernel_stats.sh <nt> <nf list> <tag>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$3
BA_NF=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$3 -- python /root/repo/tools/exp/ba_by_cameras.py $1 > /tmp/o_$3.txt 2>&1
f=$(find /tmp/prof_$3 -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$f" ]; then cp $f /root/repo/gpurun_out/ba_$3_kernel_stats.csv; head -12 $f | cut -c1-150; else tail -5 /tmp/o_$3.txt; fi
t=$(find /tmp/prof_$3 -name "*kernel_trace.csv" < /dev/null | head -1)
[ -n "$t" ] && cp $t /root/repo/gpurun_out/ba_$3_kernel_trace.csv
exit 0
