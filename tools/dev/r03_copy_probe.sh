#!/bin/bash
# Which kernels carry the host <-> device copies of the bench loop?  rocprofv3 kernel trace of the host -> host loop and of the
# device-resident loop; prints the copy / fill kernels of both (run on the GPU box through gpurun).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
for mode in host dev; do
  extra=""; [ $mode = dev ] && extra="--device-resident"
  rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-table $extra > $O/$mode.json 2> $O/$mode.err
done
find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -1 $f; grep -i "copy\|fill\|memset" $f | cut -c1-200; done
