#!/bin/bash
# Round 4 kernel iteration visit: res8 parity tests on the new kernels, A/B of the step against the previous library
# (build/old/libhowl_old.so, same box, same minute), the unfused backward of the new library, rocprof timeline of the new step.
set -u
TAG=${1:-r4b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu (${2:-res8 or ddp or engine})" ; timeout 900 python -m pytest tests -m gpu -q -x -k "${2:-res8 or ddp or engine}" 2>&1 | tee $OUT/pytest_gpu.log | tail -12
ab() {
  HOWL_HIP_LIBRARY=$1 timeout 300 python bench.py --config ${3:-c3} --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; o = r['other_kernels']
print('$2', d['value'], d['ms_per_step'], d['repeats']['ms_per_step_median'], 'pair', r['avg_launch_ms'], r['frac'], 'fwd', [(v.get('avg_launch_ms'), v.get('frac')) for k, v in o.items() if k.startswith('conv3x3')], 'loss', d['final_loss'])"
}
for i in 1 2; do
  [ -f build/old/libhowl_old.so ] && ab $PWD/build/old/libhowl_old.so old
  ab $PWD/howl_amd/libhowl_hip.so new
  HOWL_RES8_BWD_FUSED=0 ab $PWD/howl_amd/libhowl_hip.so new_unfused
done 2>&1 | tee $OUT/ab.log
for c in c1 c2; do [ -f build/old/libhowl_old.so ] && ab $PWD/build/old/libhowl_old.so old_$c $c; ab $PWD/howl_amd/libhowl_hip.so new_$c $c; done 2>&1 | tee -a $OUT/ab.log
echo "== rocprof c3" ; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o c3 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof.log 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c3_kernel_stats.csv && cut -c1-150 "$f" | head -16
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_timeline.py "$t" > $OUT/c3_step_timeline.txt && cat $OUT/c3_step_timeline.txt
rm -rf $OUT/prof
