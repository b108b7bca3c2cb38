#!/bin/bash
# round 6, visit K: the rocprof / PMC half of tools/gpu_round6_final.sh again (the first pass profiled bench.py's like_for_like leg)
set -u
OUT=gpurun_out/r6final; mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c3 c4 c5; do
  echo "== rocprof $c"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$c -o $c -- python $R/bench.py --config $c --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline --no-unfused-leg > $R/$OUT/rocprof_$c.log 2>&1
  f=$(find $R/$OUT/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/${c}_kernel_stats.csv
  anchor=logmel; [ $c = c4 ] && anchor=lstm_fwd4
  t=$(find $R/$OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" $anchor > $R/$OUT/${c}_step_timeline.txt && tail -1 $R/$OUT/${c}_step_timeline.txt
  rm -rf $R/$OUT/prof_$c
done
c=c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c4n -o c4n -- python $R/bench.py --config c4 --no-lookahead --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline --no-unfused-leg > $R/$OUT/rocprof_c4n.log 2>&1
t=$(find $R/$OUT/prof_c4n -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" lstm_fwd4 > $R/$OUT/c4_no_lookahead_step_timeline.txt && tail -1 $R/$OUT/c4_no_lookahead_step_timeline.txt
rm -rf $R/$OUT/prof_c4n
cd $R
echo "== pmc c3 at 80 mel bins" ; NUM_MELS=80 bash tools/pmc_round.sh c3 > $OUT/pmc_round_c3_m80.log 2>&1
python tools/pmc_summary.py round6_pmc_c3_m80.txt "round 6, HEAD, NUM_MELS=80 bench.py --config c3" > /dev/null 2>&1; cp profiles/round6_pmc_c3_m80.txt $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc
echo "== pmc 256 x 2 s" ; PMC_EXTRA="--seconds 2 --batch-per-gpu 256" bash tools/pmc_round.sh c3 > $OUT/pmc_round_c3_2s.log 2>&1
python tools/pmc_summary.py round6_pmc_c3_2s.txt "round 6, HEAD, bench.py --config c3 --seconds 2 --batch-per-gpu 256" > /dev/null 2>&1; cp profiles/round6_pmc_c3_2s.txt $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc
for c in c3 c4; do
  echo "== pmc $c" ; bash tools/pmc_round.sh $c > $OUT/pmc_round_$c.log 2>&1
  python tools/pmc_summary.py round6_pmc_$c.txt "round 6, HEAD, bench.py --config $c" > /dev/null 2>&1; cp profiles/round6_pmc_$c.txt $OUT/ 2>/dev/null
  rm -rf gpurun_out/pmc
done
