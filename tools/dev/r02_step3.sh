#!/bin/bash
# Bisect which earlier test file makes the first SCHP graph launch of test_gpu_parity's pipelined tests fault.
o=gpurun_out/s3; mkdir -p $o
for f in bgestimate bgonline conv green objects; do
  USC_SEGV_BACKTRACE=1 timeout -k 10 400 python -m pytest tests/test_gpu_$f.py tests/test_gpu_parity.py -q -p no:faulthandler -k "not tool and (not parity or pipelined)" > $o/$f.log 2>&1
  echo "$f + parity(pipelined): rc=$? $(tail -1 $o/$f.log | cut -c1-100)"
  grep -n -A30 "native backtrace" $o/$f.log | cut -c1-200 | head -40
done
