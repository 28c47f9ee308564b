#!/bin/bash
# per-kernel standalone durations of the pixel stages (tools/dev/pixel_times.py under rocprofv3 --kernel-trace --stats) at both sizes
R=$PWD; o=$R/gpurun_out/r5_pixel; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for sz in "1080 1920" "2160 3840"; do
  tag=$(echo $sz | tr ' ' x)
  rocprofv3 --kernel-trace --stats -d $o/$tag -o t --output-format csv -- python3 $R/tools/dev/pixel_times.py $sz > $o/$tag.txt 2> $o/$tag.err
  rm -f $o/$tag/t_kernel_trace.csv
  python3 - <<P
import csv
rows=list(csv.DictReader(open("$o/$tag/t_kernel_stats.csv")))
print("== $tag")
for r in rows:
    n=r['Name']
    if any(k in n for k in ('obj_','trimap','bgdiff','vmat','seg_p','get_fg','fillBuffer')):
        print(f"{n[:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us")
P
done
