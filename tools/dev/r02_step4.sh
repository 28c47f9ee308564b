#!/bin/bash
o=gpurun_out/s6; mkdir -p $o
USC_SEGV_BACKTRACE=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -p no:faulthandler > $o/gpu_tests.log 2>&1; echo "full suite rc=$?"
grep -n "passed\|failed" $o/gpu_tests.log | head -3; grep -n -A60 "native backtrace" $o/gpu_tests.log | cut -c1-240 | head -80
