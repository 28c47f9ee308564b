#!/bin/bash
# gpurun_out/prof_<round> (tools/dev/run_profiles.sh on the GPU box) -> the tables committed under profiles/<round>/ (ROUND=r03 by default).
# usage: tools/dev/make_profile_summaries.sh <unprofiled ms per matting forward>
set -e
ROUND=${ROUND:-r03}
P=gpurun_out/prof_$ROUND; D=profiles/$ROUND; mkdir -p $D
python tools/dev/summarize_profiles.py $P/default_trace/t_kernel_stats.csv $P/default_fetch/t_counter_collection.csv $P/default_write/t_counter_collection.csv 0 \
  "bench.py default (human-parsing seg, 12 frames per forward, + stage-3 loop, 1080p), $ROUND; PMC passes use --sync-loop (1 frame per forward)" > $D/default_kernels_and_traffic.md
python tools/dev/summarize_profiles.py $P/no_seg_trace/t_kernel_stats.csv $P/no_seg_fetch/t_counter_collection.csv $P/no_seg_write/t_counter_collection.csv 0 \
  "bench.py --no-seg (stage-3 loop only, 1080p), $ROUND" $D/pmc_conv_traffic.json > $D/stage3_kernels_and_traffic.md
cp $P/default_trace/t_kernel_stats.csv $D/bench_default_kernel_stats.csv
cp $P/no_seg_trace/t_kernel_stats.csv $D/bench_no_seg_kernel_stats.csv
cp $P/no_seg_trace.json $D/bench_no_seg_profiled_run.json
python tools/dev/summarize_mfma.py $P/no_seg_mfma/t_counter_collection.csv $P/no_seg_mfma/t_kernel_trace.csv 0 ${1:-1.50} > $D/mfma_utilisation.md
