#!/bin/bash
# tools/dev/d2h_nocu_probe.py plain and under rocprofv3 --kernel-trace --stats (how many __amd_rocclr_copyBuffer launches are left?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3t
mkdir -p $O
timeout -k 10 200 python3 $R/tools/dev/d2h_nocu_probe.py > $O/plain.txt 2> $O/plain.err || { tail -5 $O/plain.err; exit 1; }
cat $O/plain.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/dev/d2h_nocu_probe.py > $O/traced.txt 2> $O/traced.err
cat $O/traced.txt
grep -i "rocclr" $O/trace/t_kernel_stats.csv | cut -c1-120
rm -f $O/trace/t_kernel_trace.csv
