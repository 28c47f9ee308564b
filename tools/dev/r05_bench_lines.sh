#!/bin/bash
# Round-5 bench lines on the GPU box (through gpurun): every kept line carries `roofline` and `cpu_baseline`.
o=gpurun_out/final_r05; mkdir -p $o
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $o/$name.json 2> $o/$name.err || { echo "$name FAILED"; tail -15 $o/$name.err; exit 1; }; cut -c1-230 $o/$name.json; }
run bench_default
run bench_driver_form --steps 20 --warmup 5
run bench_no_seg --no-seg
run bench_4k --height 2160 --width 3840 --steps 120 --cpu-frames 2
run bench_4k_clip300_strong --height 2160 --width 3840 --clip-frames 300 --cpu-frames 2 --no-cpu-single-thread
run bench_green --mode green --steps 40
run bench_bg_online --mode bg_online --steps 40
run bench_replace --mode replace
