#!/bin/bash
# Round 5, first visit: the -m gpu suite at HEAD, the entry-point loop next to the resident-tensor step, c4 / c3 baselines on this
# box, the two-stream overlap probe.     bash tools/gpu_r5a.sh
set -u
OUT=gpurun_out/r5a
mkdir -p $OUT
export NUM_MELS=40
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest_gpu.log | tail -6
: > $OUT/bench_lines.jsonl
run() { echo "== bench $*"; timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-1500; }
run --loop entry --config c3 --steps 100 --warmup 20 --cpu-baseline-seconds 8
run --loop entry --config c1 --steps 200 --warmup 30 --no-cpu-baseline
run --no-cpu-baseline
run --config c4 --no-cpu-baseline
echo "== overlap probe"; timeout 300 python tools/c4_overlap_probe.py 2>&1 | tee $OUT/overlap_probe.log | tail -3
