#!/bin/bash
# GPU-box visit for the frontend kernel: parity tests of the frontend, log-mel timing by batch size / occupancy, s_memtime
# timeline of one workgroup, then the res8 bench line.
set -u
mkdir -p gpurun_out
export NUM_MELS=40
echo "== pytest frontend/engine/collate" ; timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_engine.py tests/test_gpu_collate.py -x -q 2>&1 | tee gpurun_out/pytest_fe.log | tail -4
echo "== probe" ; timeout 600 python tools/probe_logmel.py 2>&1 | tee gpurun_out/probe_logmel.log | grep -v "^wave 1[0-9]\|^wave [4-9]" | tail -${1:-60}
