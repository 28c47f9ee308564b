#!/bin/bash
# Round 2, first GPU call: full -m gpu suite (incl. the graph-destroy lifetime test), host->host bench lines, the serial per-layer
# kernel trace behind bench.py's roofline.
o=gpurun_out/s1; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; rc=$?; tail -5 $o/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err || { tail -20 $o/bench_default.err; exit 1; }
cut -c1-400 $o/bench_default.json
timeout -k 10 300 python bench.py --no-seg --no-cpu-baseline > $o/bench_no_seg.json 2> $o/bench_no_seg.err || { tail -20 $o/bench_no_seg.err; exit 1; }
cut -c1-400 $o/bench_no_seg.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$o/layer_trace -o lt --output-format csv -- python3 $R/tools/dev/layer_table.py run 20 > $R/$o/layers.json 2> $R/$o/layers.err || { tail -20 $R/$o/layers.err; exit 1; }
cd $R
f=$(find $o/layer_trace -name '*kernel_trace.csv' | head -1)
python tools/dev/layer_table.py table $o/layers.json $f > $o/unet_layer_table.md && tail -4 $o/unet_layer_table.md
