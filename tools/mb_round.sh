#!/bin/bash
# MobileNet iteration on the GPU box: parity tests, the c5 bench line, rocprof kernel stats.  bash tools/mb_round.sh <tag>
set -u
TAG=${1:-mb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
timeout 600 python -m pytest tests/test_gpu_mobilenet.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_mb.log
timeout 300 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_c5.log | cut -c1-420
bash tools/profile_config.sh c5 $TAG 3 > /dev/null
