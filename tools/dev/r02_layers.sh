#!/bin/bash
# usage: tools/dev/r02_layers.sh <outdir under gpurun_out>   - per-layer kernel table of the matting net only (rocprofv3 kernel trace)
o=gpurun_out/$1; mkdir -p $o
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$o/layer_trace -o lt --output-format csv -- python3 $R/tools/dev/layer_table.py run 20 > $R/$o/layers.json 2> $R/$o/layers.err || { tail -20 $R/$o/layers.err; exit 1; }
cd $R
f=$(find $o/layer_trace -name '*kernel_trace.csv' | head -1)
python tools/dev/layer_table.py table $o/layers.json $f > $o/unet_layer_table.md && grep -E "winograd|Sum over|HIP events" $o/unet_layer_table.md
