#!/bin/bash
# round 6, visit A: the CTC windows, whole-clip LSTM parity, threads, eval buffers; c4 / c3 bench lines
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
python -m pytest tests/test_gpu_lstm.py tests/test_gpu_threads.py -m gpu -x -q > $O/pytest_lstm_threads.log 2>&1; echo "lstm+threads rc=$?"; tail -5 $O/pytest_lstm_threads.log
python -m pytest tests/test_gpu_engine.py -m gpu -x -q > $O/pytest_engine.log 2>&1; echo "engine rc=$?"; tail -5 $O/pytest_engine.log
python -m pytest tests/test_gpu_res8.py -m gpu -x -q -k "eval_mode or 80_mel_bins_vs_oracle or beyond_83" > $O/pytest_res8_sel.log 2>&1; echo "res8 sel rc=$?"; tail -5 $O/pytest_res8_sel.log
python bench.py --config c4 --steps 50 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 600 $O/bench_c4.json
python bench.py --config c4 --steps 50 --no-cpu-baseline --no-lookahead > $O/bench_c4_nola.json 2>> $O/bench_c4.err; tail -c 300 $O/bench_c4_nola.json
python bench.py --steps 20 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 400 $O/bench_c3.json
