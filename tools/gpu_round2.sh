#!/bin/bash
# One GPU-box visit (round 2): parity tests (all failures listed), smoke, the bench line for every BASELINE config,
# rocprof kernel stats of the headline config.  Everything lands in gpurun_out/<tag>/.
set -u
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
lscpu | head -24 > $OUT/lscpu.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu" ; timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --durations=15 ${PYTEST_ARGS:-} 2>&1 | tee $OUT/pytest_gpu.log | tail -40
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log | tail -3
fi
echo "== bench c3" ; timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tee $OUT/bench_c3.log | tail -2
for c in ${CONFIGS:-c1 c2 c4 c5}; do
  echo "== bench $c" ; timeout 600 python bench.py --config $c --steps 20 --warmup 5 --cpu-baseline-seconds 8 2>&1 | tee $OUT/bench_$c.log | tail -2
done
echo "== rocprof c3" ; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-170 "$f" | head -32
