#!/bin/bash
# bash tools/unscreen.sh <script> <video_id> <gpu ids, e.g. 0 or 0,1,2,3> [python args]
# One process per listed GPU; with several GPUs the frames are sharded into contiguous ranges (RCCL only
# broadcasts the weights and the overall background).
script=$1
src=$2
gpu_id=${3:-0}
PY_ARGS=${@:4}
echo unscreen video ${src} on gpuid ${gpu_id}
export PYTHONPATH=./
ngpu=$(echo ${gpu_id} | tr ',' '\n' | wc -l)
if [ "${ngpu}" -gt 1 ]; then
    HIP_VISIBLE_DEVICES=${gpu_id} python -m torch.distributed.run --nnodes=1 --nproc-per-node ${ngpu} \
        --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29517} tools/unscreen/${script}.py --video_id ${src} ${PY_ARGS}
else
    HIP_VISIBLE_DEVICES=${gpu_id} python tools/unscreen/${script}.py --video_id ${src} ${PY_ARGS}
fi
echo finished video ${src} on gpuid ${gpu_id}
