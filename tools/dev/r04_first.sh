#!/bin/bash
# round 4, first GPU call: the new full-size whole-loop tests and the replace bench line
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_replace.py tests/test_gpu_bgonline.py::test_bg_online_pipeline_stage_by_stage tests/test_gpu_green.py::test_green_pipeline_stage_by_stage -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 600 python bench.py --mode replace > $OUT/bench_replace.json 2> $OUT/bench_replace.err || { tail -20 $OUT/bench_replace.err; exit 1; }
cat $OUT/bench_replace.json
