#!/bin/bash
OUT=gpurun_out/${1:-tests}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
