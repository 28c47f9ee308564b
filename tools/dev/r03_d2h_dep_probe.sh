#!/bin/bash
# tools/dev/d2h_dep_probe.py, every form, plain (timing) and under rocprofv3 --kernel-trace --stats (number of blit-kernel launches)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3u
mkdir -p $O
for m in none indep event event_nocu record_only wait_only hostwait none; do
  timeout -k 10 120 python3 $R/tools/dev/d2h_dep_probe.py $m 2> $O/$m.err | tee -a $O/plain.txt || { tail -5 $O/$m.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
for m in indep event event_nocu hostwait; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace_$m -o t --output-format csv -- python3 $R/tools/dev/d2h_dep_probe.py $m 2> $O/traced_$m.err | tee -a $O/traced.txt
  echo "   blit launches: $(grep -i rocclr_copy $O/trace_$m/t_kernel_stats.csv | cut -d, -f2,4)" | tee -a $O/traced.txt
  rm -rf $O/trace_$m
done
