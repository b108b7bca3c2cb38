#!/bin/bash
# round 6, visit E: 3 x bf16 split skeleton + accuracy; whole-clip LSTM test after the tolerance note
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
timeout 300 build/bf16x3_ubench > $O/bf16x3_ubench.txt 2>&1; echo "ubench rc=$?"; cat $O/bf16x3_ubench.txt
python -m pytest tests/test_gpu_lstm.py -m gpu -q -k "whole_clip" > $O/pytest_lstm.log 2>&1; echo "lstm rc=$?"; tail -3 $O/pytest_lstm.log
