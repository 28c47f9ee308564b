#!/bin/bash
# Round-5 profile pass on the GPU box (through gpurun): per-layer kernel table of the matting net, rocprofv3 kernel-trace --stats and the
# PMC passes of the default / --no-seg loops (tools/dev/run_profiles.sh), the 4K conv traffic, kernel stats of the replace and bg-online loops.
o=gpurun_out/final_r05; mkdir -p $o
R=$PWD
bash tools/dev/r02_layers.sh final_r05 || exit 1
ROUND=r05 bash tools/dev/run_profiles.sh > $o/prof_log.txt 2>&1; tail -3 $o/prof_log.txt
P=$R/gpurun_out/prof_r05
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $P/4k_$c -o t --output-format csv -- python3 $R/bench.py --height 2160 --width 3840 --steps 8 --warmup 0 --no-cpu-baseline --no-kernel-table --no-spin-up --sync-loop --no-seg > /dev/null 2> $P/4k_$c.err
  rm -f $P/4k_$c/t_kernel_trace.csv
  echo "4k $c done"
done
rocprofv3 --kernel-trace --stats -d $P/4k_trace -o t --output-format csv -- python3 $R/bench.py --height 2160 --width 3840 --steps 24 --warmup 4 --no-cpu-baseline --no-kernel-table --no-spin-up --no-seg > $P/4k_trace.json 2> $P/4k_trace.err
rm -f $P/4k_trace/t_kernel_trace.csv
for m in replace bg_online; do
  rocprofv3 --kernel-trace --stats -d $P/${m}_trace -o t --output-format csv -- python3 $R/bench.py --mode $m --steps 40 --warmup 4 --no-cpu-baseline > $P/${m}_trace.json 2> $P/${m}_trace.err
  rm -f $P/${m}_trace/t_kernel_trace.csv
  echo "$m trace done"
done
cd $R
python tests/parity_report.py > $o/parity_report.json 2> $o/parity_report.err; tail -c 600 $o/parity_report.json
