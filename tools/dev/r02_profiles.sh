#!/bin/bash
# Round-2 measurement + profile pass on the GPU box: bench lines (default, --no-seg, 4K, other modes), the per-layer kernel table,
# rocprofv3 kernel-trace --stats of the default and --no-seg bench runs, PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy).
o=gpurun_out/final; mkdir -p $o
bash tools/dev/r02_measure.sh final || exit 1
python bench.py --height 2160 --width 3840 --steps 120 --no-cpu-baseline --no-kernel-table > $o/bench_4k.json 2> $o/bench_4k.err || exit 1
python bench.py --mode green --steps 40 --no-cpu-baseline > $o/bench_green.json 2> $o/bench_green.err || exit 1
python bench.py --mode bg_online --steps 40 --no-cpu-baseline > $o/bench_bg_online.json 2> $o/bench_bg_online.err || exit 1
for f in bench_4k bench_green bench_bg_online; do cut -c1-140 $o/$f.json; done
python tools/dev/objects_time.py > $o/objects_time.txt 2>&1; tail -6 $o/objects_time.txt
tools/dev/run_profiles.sh > $o/prof_log.txt 2>&1; tail -3 $o/prof_log.txt
