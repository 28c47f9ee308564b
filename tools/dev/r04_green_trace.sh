#!/bin/bash
# kernel trace of the green-mode bench loop (which kernels a frame is made of)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4green; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --mode green --steps 40 --warmup 4 --no-cpu-baseline > $O/trace.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
rm -f $O/trace/t_kernel_trace.csv
python3 - <<PY
import csv, json
d = json.load(open("$O/trace.json"))
print("green (traced)", d["value"], "fps", d["ms_per_step"], "ms")
rows = list(csv.DictReader(open("$O/trace/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per frame %.3f ms" % (tot / 44 / 1e6))
for r in rows[:28]:
    print("%-64s calls %6s  avg %8.1f us  per frame %7.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 44 / 1e6))
PY
