#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/bench_decode.py --records 4000000 --rpb 60 --variants 4,8,9,10,11,3 > gpurun_out/c7_decode.txt 2>&1
timeout 300 python tools/bench_decode.py --records 2000000 --rpb 8,60 --variants 5,4,8,10,11 >> gpurun_out/c7_decode.txt 2>&1
cat gpurun_out/c7_decode.txt
