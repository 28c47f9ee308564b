#!/bin/bash
set -e
mkdir -p gpurun_out/r5_green
for lib in libunscreen_hip_regstage.so libunscreen_hip.so libunscreen_hip_regstage.so libunscreen_hip.so; do
  python tools/dev/bench_with_lib.py video_unscreen_amd/$lib --mode green --steps 40 --no-cpu-baseline > gpurun_out/r5_green/$lib.json 2> gpurun_out/r5_green/$lib.err
  python - <<P
import json
j=json.loads(open("gpurun_out/r5_green/$lib.json").read().strip().splitlines()[-1])
print("$lib", j["value"], j["ms_per_step"], j.get("roofline",{}).get("achieved"), flush=True)
P
done
