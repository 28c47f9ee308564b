#!/bin/bash
# Round 2, GPU call: the branch-stream probe (both HIP runtimes), the Winograd layer tests, then the full suite on chain graphs.
o=gpurun_out/s7; mkdir -p $o
timeout -k 10 120 tools/dev/graph_branch_stream_repro 4 16 > $o/repro_rocm72.log 2>&1; echo "repro (ROCm 7.2 runtime) rc=$? $(tail -1 $o/repro_rocm72.log)"
T=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
LD_LIBRARY_PATH=$T timeout -k 10 120 tools/dev/graph_branch_stream_repro 4 16 > $o/repro_torchlib.log 2>&1; echo "repro (torch's bundled runtime) rc=$? $(tail -1 $o/repro_torchlib.log)"
LD_LIBRARY_PATH=$T timeout -k 10 120 tools/dev/graph_branch_stream_repro 8 40 > $o/repro_torchlib_8.log 2>&1; echo "repro 8 branches (torch's runtime) rc=$? $(tail -1 $o/repro_torchlib_8.log)"
USC_SEGV_BACKTRACE=1 timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -s -p no:faulthandler -k "winograd" > $o/wino_tests.log 2>&1; echo "winograd tests rc=$?"; tail -15 $o/wino_tests.log | cut -c1-250
USC_SEGV_BACKTRACE=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -p no:faulthandler > $o/gpu_tests.log 2>&1; echo "full suite rc=$?"
tail -5 $o/gpu_tests.log | cut -c1-250; grep -n -A25 "SIGSEGV" $o/gpu_tests.log | cut -c1-200 | head -40
