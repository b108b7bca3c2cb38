#!/bin/bash
# Round 4, first GPU-box visit: the -m gpu suite at the hygiene commit, the headline bench line, the same step with the
# frontend in train mode (--vtlp), and the 2-rank control-flow check of bench.py's rccl section (gloo: two ranks on one GPU).
set -u
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | tee $OUT/pytest_gpu.log | tail -22
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log | tail -2
: > $OUT/bench_lines.jsonl
echo "== bench c3" ; timeout 600 python bench.py --cpu-baseline-seconds 6 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-400
echo "== bench c3 --vtlp" ; timeout 600 python bench.py --vtlp --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-400
for mode in overlap merged; do
echo "== bench c3 2 ranks gloo ($mode)" ; HOWL_DP_LATE=$mode HOWL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['rccl'])"
done
echo "== bench c3 strong 2 ranks gloo" ; HOWL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --global-batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | cut -c1-330
