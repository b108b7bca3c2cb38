#!/bin/bash
# Round 5: the look-ahead frontend riding in the seq-lstm forward launch -- tests, A/B, timeline; and the frontend kernel itself after its move into a header.
set -u
OUT=gpurun_out/r5j
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_frontend.py -m gpu -q 2>&1 | tail -4
: > $OUT/bench_lines.jsonl
ab() { echo "== $1"; shift; cfg=$1; shift; timeout 600 python bench.py --config $cfg --no-cpu-baseline "$@" 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); lm=(d.get('roofline') or {}).get('other_kernels',{}).get('logmel',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), d['final_loss'], 'logmel', lm.get('avg_launch_ms'))"; }
ab "c4 look-ahead" c4
ab "c4 no look-ahead" c4 --no-lookahead
ab "c4 look-ahead" c4
ab "c3 (frontend kernel after the header move)" c3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c4 -o c4 -- python $R/bench.py --config c4 --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_c4.log 2>&1
t=$(find $R/$OUT/prof_c4 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" lstm_fwd4 > $R/$OUT/c4_step_timeline.txt && cat $R/$OUT/c4_step_timeline.txt
f=$(find $R/$OUT/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/c4_kernel_stats.csv
rm -rf $R/$OUT/prof_c4
