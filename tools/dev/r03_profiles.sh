#!/bin/bash
# Round-3 measurement + profile pass on the GPU box (through gpurun): bench lines (default at 300 steps, the driver's 20-step form, --no-seg,
# 4K, the 4K 300-frame strong-scaling clip on one GPU, green, bg_online), pixel-stage durations, host-boundary ablation, the per-layer
# kernel table of the matting net, rocprofv3 kernel-trace --stats and the PMC passes (tools/dev/run_profiles.sh).
o=gpurun_out/final_r03; mkdir -p $o
bash tools/dev/r02_measure.sh final_r03 || exit 1
python bench.py --steps 20 --warmup 5 > $o/bench_driver_form.json 2> $o/bench_driver_form.err || exit 1
python bench.py --height 2160 --width 3840 --steps 120 --no-cpu-baseline --no-kernel-table > $o/bench_4k.json 2> $o/bench_4k.err || exit 1
python bench.py --height 2160 --width 3840 --clip-frames 300 --no-cpu-baseline --no-kernel-table > $o/bench_4k_clip300_strong.json 2> $o/bench_4k_clip.err || exit 1
python bench.py --mode green --steps 40 > $o/bench_green.json 2> $o/bench_green.err || exit 1
python bench.py --mode bg_online --steps 40 > $o/bench_bg_online.json 2> $o/bench_bg_online.err || exit 1
for f in bench_driver_form bench_4k bench_4k_clip300_strong bench_green bench_bg_online; do cut -c1-150 $o/$f.json; done
python tools/dev/pixel_times.py > $o/pixel_times_1080p.txt 2>&1; python tools/dev/pixel_times.py 2160 3840 > $o/pixel_times_4k.txt 2>&1
python tools/dev/hostio_ablate.py > $o/hostio_ablate.txt 2>&1; tail -11 $o/hostio_ablate.txt
python tools/dev/hostio_ablate.py --height 2160 --width 3840 --frames 120 --minimal > $o/hostio_ablate_4k.txt 2>&1; tail -4 $o/hostio_ablate_4k.txt
python tools/dev/hostio_timeline.py > $o/hostio_timeline_4k.txt 2>&1; grep "pass:" $o/hostio_timeline_4k.txt
python tools/dev/hostio_check.py 120 > $o/loops_agree_1080p.txt 2>&1; grep -c identical $o/loops_agree_1080p.txt
python tools/dev/concurrent_conv_check.py default 100 > $o/concurrent_conv_check.txt 2>&1; tail -4 $o/concurrent_conv_check.txt
python tools/dev/objects_time.py > $o/objects_time.txt 2>&1; tail -4 $o/objects_time.txt
ROUND=r03 tools/dev/run_profiles.sh > $o/prof_log.txt 2>&1; tail -3 $o/prof_log.txt
