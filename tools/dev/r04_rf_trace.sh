#!/bin/bash
# rocprofv3 --kernel-trace --stats of the Laplace solver alone on the bench's hole (tools/dev/regionfill_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/dev/regionfill_probe.py > $O/probe.txt 2> $O/probe.err
cat $O/probe.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:22]:
    print("%-60s calls %6s  avg %8.1f us  total %7.1f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -f $O/trace/t_kernel_trace.csv
