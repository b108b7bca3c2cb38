#!/bin/bash
# NOTE: HOWL_MB_PW_INTERLEAVE exists in commit 6c95866 only (measured, reverted in a5ae4e5); at HEAD the switch is ignored.
set -u
OUT=gpurun_out/r5h
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mobilenet.py -m gpu -q 2>&1 | tail -3
ab() { echo "== $1"; shift; env "$@" timeout 600 python bench.py --config c5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), d['final_loss'])"; }
ab "c5 interleaved" A=1
ab "c5 role by role" HOWL_MB_PW_INTERLEAVE=0
ab "c5 interleaved" A=1
for m in 1 0; do
  echo "== pmc c5 interleave=$m"; HOWL_MB_PW_INTERLEAVE=$m bash tools/pmc_round.sh c5 > $OUT/pmc_round_c5_$m.log 2>&1
  python tools/pmc_traffic.py round5_c5_hbm_traffic_ilv$m.txt 6 2>&1 | tail -2; cp profiles/round5_c5_hbm_traffic_ilv$m.txt $OUT/ 2>/dev/null
  rm -rf gpurun_out/pmc
done
