#!/bin/bash
# usage: tools/dev/trace_bench.sh <outdir under gpurun_out> [bench args]   - rocprofv3 kernel trace of a short bench run (device-resident loop only)
o=gpurun_out/$1; shift; mkdir -p $o
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/$o/trace -o t --output-format csv -- python3 $R/bench.py --steps 120 --warmup 24 --no-cpu-baseline --no-kernel-table --device-resident "$@" > $R/$o/bench.json 2> $R/$o/bench.err || { tail -20 $R/$o/bench.err; exit 1; }
cd $R
python tools/dev/trace_timeline.py $(find $o/trace -name '*kernel_trace.csv' | head -1)
