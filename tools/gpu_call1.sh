#!/bin/bash
# round-4 GPU call 1: new alive kernels — ubench variants, VALU rates, decode variants, alive parity tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== ubench_valu"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubv 2>/dev/null && timeout 60 /tmp/ubv
  echo "== ab_alive"; AB_FULL=tools/ubench_alive_ab_new timeout 900 bash tools/ab_alive.sh run
} > gpurun_out/c1_ab.txt 2>&1
timeout 200 python tools/bench_decode.py --records 2000000 --rpb 60,500 --variants 0,2,3,4 > gpurun_out/c1_decode.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline.py -x -q -m gpu \
  -k "alive or partitioned or collision or last_writer or contention or bit_set or config_1" > gpurun_out/c1_pytest.txt 2>&1
tail -5 gpurun_out/c1_pytest.txt
cat gpurun_out/c1_ab.txt | tail -40
