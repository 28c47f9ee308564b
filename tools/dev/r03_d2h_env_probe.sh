#!/bin/bash
# Does any runtime setting move pinned device -> host copies from the blit kernel (__amd_rocclr_copyBuffer) to an SDMA engine?
# tools/dev/d2h_batch_probe.py under rocprofv3 --kernel-trace --stats for a list of environment settings (GPU box, through gpurun).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3j
mkdir -p $O
run() {
  tag=$1; shift
  echo "=== $tag: $*"
  ( export "$@"; rocprofv3 --kernel-trace --stats -d $O/$tag -o t --output-format csv -- python3 $R/tools/dev/d2h_batch_probe.py > $O/$tag.txt 2> $O/$tag.err )
  grep -E "alone  |2 copies|1 packed" $O/$tag.txt
  grep -i "rocclr_copy" $O/$tag/t_kernel_stats.csv | cut -c1-100
  rm -f $O/$tag/t_kernel_trace.csv
}
run default USC_NOP=1
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run blit1 GPU_FORCE_BLIT_COPY_SIZE=1
run cpdma GPU_CP_DMA_COPY_SIZE=0
run pinxfer GPU_PINNED_MIN_XFER_SIZE=1 GPU_PINNED_XFER_SIZE=64
run serialize AMD_SERIALIZE_COPY=3
