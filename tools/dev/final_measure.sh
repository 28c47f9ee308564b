#!/bin/bash
# Round-end measurements on the GPU box: bench lines (default, --no-seg, 4K, other modes), per-net timings, then the rocprofv3 passes.
o=gpurun_out/final; mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench_default.err || exit 1
python bench.py --no-seg --no-cpu-baseline > $o/bench_no_seg.json 2> $o/bench_no_seg.err || exit 1
python bench.py --height 2160 --width 3840 --steps 120 --no-cpu-baseline > $o/bench_4k.json 2> $o/bench_4k.err || exit 1
python bench.py --mode green --steps 40 --no-cpu-baseline > $o/bench_green.json 2> $o/bench_green.err || exit 1
python bench.py --mode bg_online --steps 40 --no-cpu-baseline > $o/bench_bg_online.json 2> $o/bench_bg_online.err || exit 1
python tools/dev/profile_modes.py > $o/modes.txt 2>&1 || exit 1
python tools/dev/profile_segagent.py > $o/segagent.txt 2>&1 || exit 1
python tools/dev/profile_bgonline.py > $o/bgonline.txt 2>&1 || exit 1
python tools/dev/profile_convs.py > $o/unet_layers.txt 2>&1 || exit 1
B=4 PLAN=4 python tools/dev/profile_schp.py > $o/schp_layers.txt 2>&1 || exit 1
B=1 PLAN=4 python tools/dev/profile_schp.py > $o/schp_layers_b1.txt 2>&1 || exit 1
python tools/dev/profile_segments.py > $o/schp_segments.txt 2>&1 || exit 1
for f in bench_default bench_no_seg bench_4k bench_green bench_bg_online; do cut -c1-140 $o/$f.json; done
tail -3 $o/modes.txt; tail -4 $o/segagent.txt; tail -12 $o/bgonline.txt; tail -2 $o/unet_layers.txt; tail -1 $o/schp_layers.txt $o/schp_layers_b1.txt; tail -2 $o/schp_segments.txt
tools/dev/run_profiles.sh > $o/prof_log.txt 2>&1; tail -2 $o/prof_log.txt
