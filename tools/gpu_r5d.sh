#!/bin/bash
# NOTE: HOWL_LSTM_RIDE_X (the projection riders of the forward recurrence) exists in commits abe4f8a..08d1d79 only; at HEAD the switch is ignored.
# Round 5, fourth visit: c4 with the input projection riding in the forward recurrence's launch + AdamW in the fold: A/B + timeline.
set -u
OUT=gpurun_out/r5d
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 1500 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_engine.py -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -4
: > $OUT/bench_lines.jsonl
ab() { echo "== c4 $1"; shift; env "$@" timeout 600 python bench.py --config c4 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), r.get('ms_per_step_min'), r.get('ms_per_step_max'), d['final_loss'])"; }
ab "default (riders + fold AdamW)" A=1
ab "projection fused into the step" HOWL_LSTM_RIDE_X=0
ab "separate AdamW launch" HOWL_NO_FOLD_ADAMW=1
ab "both off" HOWL_LSTM_RIDE_X=0 HOWL_NO_FOLD_ADAMW=1
ab "default again" A=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c4 -o c4 -- python $R/bench.py --config c4 --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_c4.log 2>&1
f=$(find $R/$OUT/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/c4_kernel_stats.csv
t=$(find $R/$OUT/prof_c4 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" > $R/$OUT/c4_step_timeline.txt && cat $R/$OUT/c4_step_timeline.txt
rm -rf $R/$OUT/prof_c4
