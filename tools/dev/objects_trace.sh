#!/bin/bash
# usage: tools/dev/objects_trace.sh <outdir under gpurun_out>   - per-kernel standalone durations of the device object filter (rocprofv3 kernel trace)
o=gpurun_out/$1; mkdir -p $o
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$o/obj_trace -o t --output-format csv -- python3 $R/tools/dev/objects_time.py > $R/$o/objects_time.txt 2>&1 || { tail -20 $R/$o/objects_time.txt; exit 1; }
cd $R
grep -v amdgpu $o/objects_time.txt | tail -3
python - <<PY
import csv, glob, collections, re
f = glob.glob("$o/obj_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
# the first 23 filter calls are the clean silhouette (3 warm + 20 timed): take calls 4..23 of each kernel
by = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    by[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for n, v in by.items():
    if n.startswith("obj_") or "fillBuffer" in n:
        w = v[3:23]
        print(f"  {n:28s} {sum(w) / len(w) / 1e3:7.2f} us (clean silhouette, standalone)")
        tot += sum(w) / len(w) / 1e3
print(f"  sum {tot:.1f} us")
PY
