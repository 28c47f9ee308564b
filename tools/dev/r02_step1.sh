#!/bin/bash
# Round 2, first GPU call: graph-destroy probe, host->host bench lines, the serial per-layer kernel trace behind bench.py's roofline,
# then the full -m gpu suite with the native SIGSEGV backtrace armed.
o=gpurun_out/s1; mkdir -p $o
timeout -k 10 120 tools/dev/graph_destroy_repro 200 > $o/graph_repro.log 2>&1; echo "repro rc=$?"; tail -2 $o/graph_repro.log
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err || { tail -20 $o/bench_default.err; exit 1; }
cut -c1-600 $o/bench_default.json
timeout -k 10 300 python bench.py --no-seg --no-cpu-baseline > $o/bench_no_seg.json 2> $o/bench_no_seg.err || { tail -20 $o/bench_no_seg.err; exit 1; }
cut -c1-600 $o/bench_no_seg.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$o/layer_trace -o lt --output-format csv -- python3 $R/tools/dev/layer_table.py run 20 > $R/$o/layers.json 2> $R/$o/layers.err || { tail -20 $R/$o/layers.err; exit 1; }
cd $R
f=$(find $o/layer_trace -name '*kernel_trace.csv' | head -1)
python tools/dev/layer_table.py table $o/layers.json $f > $o/unet_layer_table.md && tail -4 $o/unet_layer_table.md
USC_SEGV_BACKTRACE=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; rc=$?; grep -n "passed\|failed\|Fatal" $o/gpu_tests.log | head; grep -n -A40 "native backtrace" $o/gpu_tests.log | cut -c1-200 | head -60
exit $rc
