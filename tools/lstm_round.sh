#!/bin/bash
# GPU-box visit while iterating on the seq-lstm step (c4): MFMA broadcast probe, LSTM parity tests, bench line, rocprof kernel
# stats + one-step timeline.   bash tools/lstm_round.sh [tag]
set -u
TAG=${1:-lstm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma4x4_probe.hip -o /tmp/mfma4x4_probe 2>/dev/null && timeout 60 /tmp/mfma4x4_probe > $OUT/mfma_probe.log 2>&1
grep "cbsz" $OUT/mfma_probe.log
echo "== pytest lstm / engine" ; timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -6
echo "== bench c4" ; timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | tee $OUT/bench_c4.log | tail -1 | cut -c1-600
bash tools/profile_config.sh c4 $TAG 30
python tools/step_timeline.py $(find $OUT/prof_c4 -name '*kernel_trace.csv' | head -1) > $OUT/c4_step_timeline.txt 2>&1; cat $OUT/c4_step_timeline.txt
