#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 200 python tools/bench_decode.py --records 2000000 --rpb 8,60,500 --variants 0 > gpurun_out/c5_decode.txt 2>&1
timeout 200 python tools/bench_decode.py --records 4000000 --rpb 60 --variants 0,3 >> gpurun_out/c5_decode.txt 2>&1
cat gpurun_out/c5_decode.txt
timeout 900 python -m pytest tests/test_kafka_decode.py tests/test_gpu_parity.py -x -q -m gpu -k "decode or corrupt or segment or blob or both_handlers_in_one_pass_equals" > gpurun_out/c5_pytest.txt 2>&1
tail -5 gpurun_out/c5_pytest.txt
