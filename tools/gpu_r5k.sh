#!/bin/bash
# round 5: res8 / frontend at 80 mel bins on the device + the default bench line (unchanged kernels) beside it
OUT=gpurun_out/${1:-r5k}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_res8.py tests/test_gpu_frontend.py -x -q -m gpu -k "80_mel" > $OUT/pytest_80.log 2>&1
tail -15 $OUT/pytest_80.log
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > $OUT/bench_c3.jsonl 2> $OUT/bench_c3.err
tail -c 600 $OUT/bench_c3.jsonl
NUM_MELS=80 timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > $OUT/bench_c3_m80.jsonl 2> $OUT/bench_c3_m80.err
tail -c 1500 $OUT/bench_c3_m80.jsonl; tail -5 $OUT/bench_c3_m80.err
NUM_MELS=80 timeout 300 python bench.py --no-cpu-baseline --config c1 --steps 50 --warmup 10 > $OUT/bench_c1_m80.jsonl 2> $OUT/bench_c1_m80.err
tail -c 400 $OUT/bench_c1_m80.jsonl
