#!/bin/bash
# PMC passes over single conv layers of the matting net (stall breakdown).  Usage: tools/dev/pmc_layer.sh <layer> [<layer> ...]
set -e
repo=$PWD
out=$repo/gpurun_out/pmc_layer; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/L${L}_p$i -- python3 $repo/tools/dev/one_layer.py $L 10 > $out/L${L}_p$i.log 2>&1
    echo "layer $L pass $i done"
  done
done
