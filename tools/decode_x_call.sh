#!/bin/bash
# One gpurun call: the experiment switches of kafka_decode_coop (csrc/kta_decode_coop.h: DX_*) side by side, from the library
# built with -DKTA_DECODE_EXPERIMENTS (tools/build_variant.sh x -DKTA_DECODE_EXPERIMENTS), then SQ counters of the plain kernel,
# its ablations and the candidates.   variant 1000 + X: <4, 3 KiB, 16>, 2000 + X: <2, 8 KiB, 32>
#   gpurun --timeout 420 -- bash tools/decode_x_call.sh
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/decx
mkdir -p $OUT
cd $ROOT
LIB=kafka_topic_analyzer_amd/libkta_hip.so.x
X="0 1 2 3 4 5 7 8 9 16 32 64 65 69 128 129 133 135 256 257 259 263"
V4=$(for x in $X; do printf "%d," $((1000 + x)); done | sed 's/,$//')
V2=$(for x in $X; do printf "%d," $((2000 + x)); done | sed 's/,$//')
date +%T
timeout 150 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 60 --variants $V4,$V2 --sweeps 2 --reps 5 > $OUT/t_rpb60_2m.jsonl 2> $OUT/t_rpb60_2m.err
date +%T
timeout 120 python tools/bench_decode.py --lib $LIB --records 4000000 --rpb 60 --variants $V4 --sweeps 2 --reps 5 > $OUT/t_rpb60_4m.jsonl 2> $OUT/t_rpb60_4m.err
date +%T
timeout 120 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 500 --variants $V2,1000,1003,1129 --sweeps 2 --reps 5 > $OUT/t_rpb500_2m.jsonl 2> $OUT/t_rpb500_2m.err
date +%T
timeout 120 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 8 --variants $V4 --sweeps 2 --reps 5 > $OUT/t_rpb8_2m.jsonl 2> $OUT/t_rpb8_2m.err
date +%T
CMD="python $ROOT/tools/bench_decode.py --lib $ROOT/$LIB --records 4000000 --rpb 60 --variants 1000,1001,1003,1008,1016,1032,1128,1256,2000 --reps 3"
timeout 150 bash tools/rocprof_sq.sh decx_sq -- $CMD
date +%T
timeout 150 bash tools/rocprof_pmc.sh decx_sq2 "SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" -- $CMD
date +%T
wc -l $OUT/*.jsonl
tail -3 $OUT/*.err
