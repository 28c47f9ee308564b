#!/bin/bash
# Round 2, GPU call 2: locate the graph-destroy segfault (native backtrace), and see how the host copies overlap the compute.
o=gpurun_out/s2; mkdir -p $o
USC_SEGV_BACKTRACE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -p no:faulthandler > $o/parity_alone.log 2>&1; echo "parity alone rc=$?"
grep -n "passed\|failed" $o/parity_alone.log | head -3; grep -n -A30 "native backtrace" $o/parity_alone.log | cut -c1-220 | head -50
USC_SEGV_BACKTRACE=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:faulthandler > $o/gpu_tests.log 2>&1; echo "full suite rc=$?"
grep -n "passed\|failed" $o/gpu_tests.log | head -3; grep -n -A40 "native backtrace" $o/gpu_tests.log | cut -c1-220 | head -60
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$o/hostio_trace -o t --output-format csv -- python3 $R/bench.py --no-seg --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-table > $R/$o/hostio_bench.json 2> $R/$o/hostio_bench.err
cd $R; cut -c1-200 $o/hostio_bench.json; ls $o/hostio_trace | head; 
