#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline.py tests/test_gpu_dist.py tests/test_kafka_decode.py -x -q -m gpu \
  -k "both_handlers or golden or random_stream or ragged or config_1 or test_double or decode_matches or corrupt or reset_and" > gpurun_out/c4_pytest.txt 2>&1
tail -15 gpurun_out/c4_pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hostfed > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
tail -3 gpurun_out/c4_bench.err
python __graft_entry__.py --smoke 2>&1 | tail -2
