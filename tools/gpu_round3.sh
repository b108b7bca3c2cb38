#!/bin/bash
# One GPU-box visit (round 3): the whole -m gpu suite, smoke, the bench line of every configuration (+ the evaluation line, the
# small batches), rocprofv3 kernel stats + step timeline of the headline configuration, PMC passes (headline step, log-mel loop).
# Everything lands in gpurun_out/<tag>/.      bash tools/gpu_round3.sh [tag]
set -u
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu" ; timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --durations=10 2>&1 | tee $OUT/pytest_gpu.log | tail -25
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log | tail -3
fi
echo "== bench c3" ; timeout 600 python bench.py 2>&1 | tee $OUT/bench_c3.log | tail -1 | cut -c1-600
: > $OUT/bench_lines.jsonl
tail -1 $OUT/bench_c3.log >> $OUT/bench_lines.jsonl
for a in "--config c1" "--config c2" "--config c4" "--config c5" "--config eval" "--config c1 --batch-per-gpu 16" "--config c1 --batch-per-gpu 1"; do
  echo "== bench $a" ; timeout 600 python bench.py $a --cpu-baseline-seconds 6 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-330
done
echo "== rocprof c3" ; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o c3 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof.log 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-170 "$f" | head -30
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_timeline.py "$t" > $OUT/c3_step_timeline.txt && tail -1 $OUT/c3_step_timeline.txt
if [ "${SKIP_PMC:-0}" != "1" ]; then
echo "== pmc c3" ; bash tools/pmc_round.sh c3 > $OUT/pmc_round.log 2>&1; tail -3 $OUT/pmc_round.log
echo "== logmel profile" ; bash tools/fe_prof.sh > $OUT/fe_prof.log 2>&1; tail -12 $OUT/fe_prof.log | cut -c1-700
fi
