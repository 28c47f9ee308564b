#!/bin/bash
# bg-online bench line (no CPU baseline) + a kernel trace of the same loop: ms per Laplace solve and the solver kernels' averages
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4bg; mkdir -p $O
cd $R && timeout -k 10 300 python bench.py --mode bg_online --steps 40 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("bg_online", d["value"], "fps", d["ms_per_step"], "ms;", {k: d["roofline_fill"][k] for k in ("iterations", "ms_per_solve", "frac")})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --mode bg_online --steps 40 --warmup 4 --no-cpu-baseline > $O/trace.json 2> $O/trace.err
rm -f $O/trace/t_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace/t_kernel_stats.csv")))
for r in rows:
    if r["Name"].startswith(("mg_", "rf_", "void mg_", "void rf_")):
        print("%-60s calls %6s  avg %8.1f us  total %7.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
