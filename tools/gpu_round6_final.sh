#!/bin/bash
# Round 6 evidence visit at HEAD: the whole -m gpu suite, smoke, bench lines of every configuration (+ --vtlp, evaluation, small
# batches, the entry-point loops, 2-rank gloo control flow), rocprofv3 kernel stats + step timelines for c1..c5, PMC passes (c3, c4).
# Everything lands in gpurun_out/<tag>/.      bash tools/gpu_round6_final.sh [tag]
set -u
TAG=${1:-r6final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tee $OUT/pytest_gpu.log | tail -16
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log | tail -2
: > $OUT/bench_lines.jsonl
echo "== bench c3" ; timeout 600 python bench.py 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-300
for a in "--vtlp --no-cpu-baseline" "--config c1" "--config c2" "--config c4" "--config c4 --no-lookahead --no-cpu-baseline" "--config c5" "--config eval" "--config c1 --batch-per-gpu 16 --no-cpu-baseline" "--config c1 --batch-per-gpu 1 --no-cpu-baseline" "--loop entry --config c3 --steps 100 --warmup 20" "--loop entry --config c1 --steps 200 --warmup 30 --no-cpu-baseline" "--loop entry --config c2 --steps 200 --warmup 30 --no-cpu-baseline"; do
  echo "== bench $a" ; timeout 600 python bench.py $a --cpu-baseline-seconds 6 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-260
done
echo "== bench c3 at 80 mel bins (stock NUM_MELS)" ; NUM_MELS=80 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-260
echo "== bench 256 x 2 s (161 frames: row strips)" ; timeout 600 python bench.py --seconds 2 --batch-per-gpu 256 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-260
echo "== bench 256 x 2 s at 80 mel bins" ; NUM_MELS=80 timeout 600 python bench.py --seconds 2 --batch-per-gpu 256 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-260
for mode in overlap merged; do
  echo "== bench c3, 2 ranks on one GPU over gloo ($mode): control flow of the rccl section" ; HOWL_DP_LATE=$mode HOWL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --prewarm 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c3 c4 c5; do
  echo "== rocprof $c"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$c -o $c -- python $R/bench.py --config $c --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline --no-unfused-leg > $R/$OUT/rocprof_$c.log 2>&1
  f=$(find $R/$OUT/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/${c}_kernel_stats.csv
  anchor=logmel; [ $c = c4 ] && anchor=lstm_fwd4      # c4: the frontend rides in the forward recurrence's launch, which opens the step
  t=$(find $R/$OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" $anchor > $R/$OUT/${c}_step_timeline.txt && tail -1 $R/$OUT/${c}_step_timeline.txt
  rm -rf $R/$OUT/prof_$c
done
cd $R
# whole-clip CTC: the sequence objective as the reference batches it (64 x 4 s clips: 318 frames, three windows of the CTC kernel)
echo "== bench c4 on 4-s clips, batch 64" ; timeout 600 python bench.py --config c4 --seconds 4 --batch-per-gpu 64 --no-cpu-baseline --no-unfused-leg 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-260
if [ "${SKIP_PMC:-0}" != "1" ]; then
  # the strip instances (these two summaries were cited and not tracked)
  echo "== pmc c3 at 80 mel bins" ; NUM_MELS=80 bash tools/pmc_round.sh c3 > $OUT/pmc_round_c3_m80.log 2>&1
  python tools/pmc_summary.py round6_pmc_c3_m80.txt "round 6, HEAD, NUM_MELS=80 bench.py --config c3" > /dev/null 2>&1; cp profiles/round6_pmc_c3_m80.txt $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc
  echo "== pmc 256 x 2 s" ; PMC_EXTRA="--seconds 2 --batch-per-gpu 256" bash tools/pmc_round.sh c3 > $OUT/pmc_round_c3_2s.log 2>&1
  python tools/pmc_summary.py round6_pmc_c3_2s.txt "round 6, HEAD, bench.py --config c3 --seconds 2 --batch-per-gpu 256" > /dev/null 2>&1; cp profiles/round6_pmc_c3_2s.txt $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc
  for c in c3 c4; do
    echo "== pmc $c" ; bash tools/pmc_round.sh $c > $OUT/pmc_round_$c.log 2>&1; tail -2 $OUT/pmc_round_$c.log
    python tools/pmc_summary.py round6_pmc_$c.txt "round 6, HEAD, bench.py --config $c" > /dev/null 2>&1; cp profiles/round6_pmc_$c.txt $OUT/ 2>/dev/null
    rm -rf gpurun_out/pmc
  done
fi
