#!/bin/bash
# every admissible direct plan of every layer shape of the networks, timed (tools/dev/tune_convs.py) - the data the cost model's constants are checked against
set -e
mkdir -p gpurun_out/r5_tune/b12 gpurun_out/r5_tune/b4
NETS=unet,schp,deeplab,stm SCHP_B=12 python tools/dev/tune_convs.py gpurun_out/r5_tune/b12 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_tune/b12/log.txt
NETS=schp SCHP_B=4 python tools/dev/tune_convs.py gpurun_out/r5_tune/b4 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_tune/b4/log.txt
tail -1 gpurun_out/r5_tune/b12/log.txt; tail -1 gpurun_out/r5_tune/b4/log.txt
