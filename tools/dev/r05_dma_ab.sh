#!/bin/bash
# round 5: the direct kernel's A operand through LDS-DMA (shipped) against the register-staged kernel of rounds 1 - 4 (libunscreen_hip_regstage.so,
# built from the commit before): whole networks, per layer shape of the parsing net, then the bench loop
set -e
mkdir -p gpurun_out/r5_dma
python tools/dev/ab_nets.py video_unscreen_amd/libunscreen_hip_regstage.so video_unscreen_amd/libunscreen_hip.so video_unscreen_amd/libunscreen_hip_regstage.so video_unscreen_amd/libunscreen_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_dma/ab_nets.txt
B=12 PLAN=12 USC_LIB=video_unscreen_amd/libunscreen_hip_regstage.so python tools/dev/profile_schp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_dma/schp_b12_regstage.txt
B=12 PLAN=12 python tools/dev/profile_schp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_dma/schp_b12_dma.txt
tail -1 gpurun_out/r5_dma/schp_b12_regstage.txt; tail -1 gpurun_out/r5_dma/schp_b12_dma.txt
for lib in libunscreen_hip_regstage.so libunscreen_hip.so libunscreen_hip_regstage.so libunscreen_hip.so; do
  python tools/dev/bench_with_lib.py video_unscreen_amd/$lib --no-cpu-baseline --no-kernel-table --no-interleave > gpurun_out/r5_dma/bench_$lib.json 2> gpurun_out/r5_dma/bench_$lib.err
  python - <<P
import json
j=json.loads(open("gpurun_out/r5_dma/bench_$lib.json").read().strip().splitlines()[-1])
print("$lib", j["value"], j["ms_per_step"], j["device_resident"]["value"], j["self_check"], flush=True)
P
done
