#!/bin/bash
# gpurun_out/final_r05 + gpurun_out/prof_r05 (r05_bench_lines.sh, r05_profiles.sh on the GPU box) -> profiles/r05 (tracked).
# usage: tools/dev/r05_collect.sh <unprofiled ms per matting forward, from bench_default.json roofline.ms_per_forward>
set -e
o=gpurun_out/final_r05; P=gpurun_out/prof_r05; D=profiles/r05
for f in bench_default bench_driver_form bench_no_seg bench_4k bench_4k_clip300_strong bench_green bench_bg_online bench_replace; do cp $o/$f.json $D/$f.json; done
cp $o/unet_layer_table.md $D/unet_layer_table.md
cp $o/parity_report.json $D/parity_report.json
ROUND=r05 bash tools/dev/make_profile_summaries.sh ${1:-1.216}
python tools/dev/summarize_profiles.py $P/4k_trace/t_kernel_stats.csv $P/4k_FETCH_SIZE/t_counter_collection.csv $P/4k_WRITE_SIZE/t_counter_collection.csv 0 \
  "bench.py --no-seg --height 2160 --width 3840 (stage-3 loop only), r05" $D/pmc_conv_traffic_3840x2160.json > $D/stage3_kernels_and_traffic_4k.md
cp $P/4k_trace/t_kernel_stats.csv $D/bench_4k_no_seg_kernel_stats.csv
cp $P/replace_trace/t_kernel_stats.csv $D/bench_replace_kernel_stats.csv
cp $P/bg_online_trace/t_kernel_stats.csv $D/bench_bg_online_kernel_stats.csv
