#!/bin/bash
# usage: tools/dev/r02_measure.sh <outdir under gpurun_out> [tests]     - GPU suite (optional), bench lines, per-layer kernel table
o=gpurun_out/$1; mkdir -p $o
if [ "$2" = "tests" ]; then
  USC_SEGV_BACKTRACE=1 timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s -p no:faulthandler > $o/gpu_tests.log 2>&1; rc=$?
  echo "gpu suite rc=$rc: $(tail -1 $o/gpu_tests.log | cut -c1-120)"; grep -n -A25 "SIGSEGV\|Error\|FAILED" $o/gpu_tests.log | cut -c1-220 | head -50
  [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err || { tail -20 $o/bench_default.err; exit 1; }
python - <<PY
import json
z=json.load(open("$o/bench_default.json"))
print("default:", z["value"], "fps; device_resident", z.get("device_resident",{}).get("value"), "host_filter", z.get("host_object_filter",{}).get("value"), "roofline", z["roofline"]["achieved"], z["roofline"]["frac"], z["roofline"]["ms_per_forward"], z["roofline"]["ms_per_forward_graph"])
PY
timeout -k 10 300 python bench.py --no-seg --no-cpu-baseline --no-kernel-table > $o/bench_no_seg.json 2> $o/bench_no_seg.err || { tail -20 $o/bench_no_seg.err; exit 1; }
python - <<PY
import json
z=json.load(open("$o/bench_no_seg.json"))
print("no-seg:", z["value"], "fps; device_resident", z.get("device_resident",{}).get("value"), "host_filter", z.get("host_object_filter",{}).get("value"))
PY
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$o/layer_trace -o lt --output-format csv -- python3 $R/tools/dev/layer_table.py run 20 > $R/$o/layers.json 2> $R/$o/layers.err || { tail -20 $R/$o/layers.err; exit 1; }
cd $R
f=$(find $o/layer_trace -name '*kernel_trace.csv' | head -1)
python tools/dev/layer_table.py table $o/layers.json $f > $o/unet_layer_table.md && tail -4 $o/unet_layer_table.md
rm -rf $o/layer_trace
