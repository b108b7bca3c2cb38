#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export NUM_MELS=40
rocm-smi --showclocks --showproductname > gpurun_out/rocm_smi.txt 2>&1
lscpu | head -20 > gpurun_out/lscpu.txt 2>&1
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log | tail -5
echo "== bench" ; timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tee gpurun_out/bench.log | tail -3
echo "== rocprof" ; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
echo "== kernel scaling"; timeout 300 python tools/kernel_scaling.py 2>&1 | tee gpurun_out/kernel_scaling.log | tail -6
echo "== other configs"; timeout 300 python tools/bench_configs.py 2>&1 | tee gpurun_out/bench_configs.log | tail -6
