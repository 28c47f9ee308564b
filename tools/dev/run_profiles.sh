#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace + two PMC passes of bench.py (default and --no-seg), outputs under
# gpurun_out/prof_<round>/ (ROUND=r03 by default).  Summaries are made afterwards with tools/dev/summarize_profiles.py and committed under profiles/<round>/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r03}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in default no_seg; do
  extra=""; [ $mode = no_seg ] && extra="--no-seg"
  rocprofv3 --kernel-trace --stats -d $OUT/${mode}_trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-table --no-spin-up $extra > $OUT/${mode}_trace.json 2> $OUT/${mode}_trace.err
  echo "trace $mode done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${mode}_fetch -o t --output-format csv -- python3 $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-kernel-table --no-spin-up --sync-loop $extra > /dev/null 2> $OUT/${mode}_fetch.err
  echo "fetch $mode done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${mode}_write -o t --output-format csv -- python3 $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-kernel-table --no-spin-up --sync-loop $extra > /dev/null 2> $OUT/${mode}_write.err
  echo "write $mode done"
done
# (only the stats / counter tables travel back: the raw kernel traces of these passes are tens of MB)
rm -f $OUT/*_trace/t_kernel_trace.csv $OUT/*_fetch/t_kernel_trace.csv $OUT/*_write/t_kernel_trace.csv
ls -R $OUT | head -40
# MFMA utilisation pass (SQ + GRBM counters, stage-3 loop only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/no_seg_mfma -o t --output-format csv -- python3 $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-kernel-table --no-spin-up --sync-loop --no-seg > /dev/null 2> $OUT/no_seg_mfma.err
echo "mfma pass done"
