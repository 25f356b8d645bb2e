#!/bin/bash
# round-4 GPU call 2: store experiments of the partition kernel, the two-ahead sweep, new bench rows, new tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB_FULL=tools/ubench_alive_ab_new timeout 600 bash tools/ab_alive.sh run > gpurun_out/c2_ab.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-hostfed > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_report_cli.py -x -q -m gpu \
  -k "adversarial or test_double or eight_ranks" > gpurun_out/c2_pytest.txt 2>&1
tail -5 gpurun_out/c2_pytest.txt; tail -3 gpurun_out/c2_bench.err; cat gpurun_out/c2_ab.txt
