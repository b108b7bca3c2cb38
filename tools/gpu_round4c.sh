#!/bin/bash
# Round 4 kernel iteration visit: res8 parity tests, variants A/B (tools/variants4.py run <names>), rocprof timeline.
set -u
TAG=${1:-r4f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu (${2:-res8 or ddp or engine})" ; timeout 900 python -m pytest tests -m gpu -q -x -k "${2:-res8 or ddp or engine}" 2>&1 | tee $OUT/pytest_gpu.log | tail -12
echo "== variants ${3:-}" ; python tools/variants4.py run ${3:-base} 2>&1 | tee $OUT/variants.log
echo "== rocprof c3" ; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o c3 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof.log 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c3_kernel_stats.csv
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_timeline.py "$t" > $OUT/c3_step_timeline.txt && cat $OUT/c3_step_timeline.txt
rm -rf $OUT/prof
