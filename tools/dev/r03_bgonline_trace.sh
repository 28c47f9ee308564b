#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bg online loop: where do the 12 ms of the Laplace solver go (kernel time vs launch chain)?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --mode bg_online --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:28]:
    print("%-70s calls %6s  avg %8.1f us  total %7.1f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cut -c1-300 $O/bench.json
rm -f $O/trace/t_kernel_trace.csv
