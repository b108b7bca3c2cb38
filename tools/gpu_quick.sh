#!/bin/bash
# Short GPU-box visit while iterating on a kernel: res8 parity tests, bench (no CPU baseline), rocprof kernel stats.
set -u
mkdir -p gpurun_out
export NUM_MELS=40
echo "== pytest -m gpu (${1:-all})" ; timeout 600 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} 2>&1 | tee gpurun_out/pytest_gpu.log | tail -8
echo "== bench" ; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tee gpurun_out/bench.log | tail -2
echo "== rocprof" ; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -14
