#!/bin/bash
# Round 5, second visit: res8 / lstm GPU tests at HEAD; c4 A/B (dual weight-gradient job, side lane) on one box + timeline.
set -u
OUT=gpurun_out/r5b
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 1500 python -m pytest tests/test_gpu_res8.py tests/test_gpu_lstm.py -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -8
: > $OUT/bench_lines.jsonl
ab() { echo "== c4 $1"; shift; env "$@" timeout 600 python bench.py --config c4 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('repeats',{}))"; }
ab "default (dual job + side lane)" A=1
ab "no side lane" HOWL_NO_SIDE_STREAM=1
ab "no dual job" HOWL_WGRAD_NO_DUAL=1
ab "neither (round 4 structure)" HOWL_NO_SIDE_STREAM=1 HOWL_WGRAD_NO_DUAL=1
ab "default again" A=1
echo "== c3"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('repeats',{}))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c4 -o c4 -- python $R/bench.py --config c4 --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_c4.log 2>&1
f=$(find $R/$OUT/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/c4_kernel_stats.csv
t=$(find $R/$OUT/prof_c4 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" > $R/$OUT/c4_step_timeline.txt && cat $R/$OUT/c4_step_timeline.txt
rm -rf $R/$OUT/prof_c4
