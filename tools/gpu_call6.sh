#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/rocprof_sq.sh dec60 -- python /root/repo/tools/bench_decode.py --records 4000000 --rpb 60,500 --variants 0 --reps 3
bash tools/rocprof_pmc.sh dec60b "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CYCLES" -- python /root/repo/tools/bench_decode.py --records 4000000 --rpb 60,500 --variants 0 --reps 3
grep "kafka_decode_coop" gpurun_out/prof_dec60/sq.txt gpurun_out/prof_dec60b/pmc.txt
