#!/bin/bash
# Do pinned-host <-> device copies run on the SDMA engines or as blit kernels on the CUs, and which runtime setting changes that?
# tools/dev/copy_engine_probe.py under several environment settings, each under rocprofv3 --kernel-trace --stats (the copy kernels show
# up as __amd_rocclr_copyBuffer).  Run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
run() {
  tag=$1; shift
  echo "=== $tag: $*"
  ( export "$@"; rocprofv3 --kernel-trace --stats -d $O/$tag -o t --output-format csv -- python3 $R/tools/dev/copy_engine_probe.py > $O/$tag.txt 2> $O/$tag.err )
  tail -3 $O/$tag.txt
  grep -i "rocclr\|copy" $O/$tag/t_kernel_stats.csv | cut -c1-140
  rm -f $O/$tag/t_kernel_trace.csv
}
run default USC_NOP=1
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
run blitwg4 DEBUG_CLR_LIMIT_BLIT_WG=4
run engine2 GPU_BLIT_ENGINE_TYPE=2
