#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB_FULL=tools/ubench_alive_ab_new timeout 600 bash tools/ab_alive.sh run > gpurun_out/c3_ab.txt 2>&1
timeout 200 python tools/bench_decode.py --records 2000000 --rpb 500,2000 --variants 0,3,6,7 > gpurun_out/c3_decode.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-hostfed > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
cat gpurun_out/c3_ab.txt gpurun_out/c3_decode.txt; tail -3 gpurun_out/c3_bench.err
