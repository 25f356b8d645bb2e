#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB_FULL=none timeout 300 bash tools/ab_alive.sh run > gpurun_out/c8_ab.txt 2>&1
timeout 300 python tools/bench_decode.py --records 2000000 --rpb 8,60,500 --variants 0 > gpurun_out/c8_decode.txt 2>&1
timeout 600 python -m pytest tests/test_kafka_decode.py -x -q -m gpu -k "decode_matches or corrupt" > gpurun_out/c8_pytest.txt 2>&1
cat gpurun_out/c8_ab.txt; grep "^{" gpurun_out/c8_decode.txt | cut -c1-200; tail -3 gpurun_out/c8_pytest.txt
