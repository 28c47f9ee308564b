#!/bin/bash
# Sample the GPU clock / power while bench.py runs (is the MFMA peak we price against reachable under sustained load?).
out=gpurun_out/clock_probe.txt; : > $out
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-kernel-table > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
pid=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use|busy" >> $out
  echo "--" >> $out
  sleep 0.7
done
wait $pid
cat $out | head -60
cut -c1-160 gpurun_out/clock_probe_bench.json
