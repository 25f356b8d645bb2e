#!/bin/bash
# One gpurun call: the experiment switches of kafka_decode_coop (csrc/kta_decode_coop.h: DX_*) side by side, from the library
# built with -DKTA_DECODE_EXPERIMENTS (tools/build_variant.sh x -DKTA_DECODE_EXPERIMENTS).
#   variant 1000 + X: <4, 3 KiB, 16>, 2000 + X: <2, 8 KiB, 32>
#   gpurun --timeout 300 -- bash tools/decode_x_call.sh
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/decx3
mkdir -p $OUT
cd $ROOT
LIB=kafka_topic_analyzer_amd/libkta_hip.so.x
X="0 3 259 1024 1027 1283 1287"
V4=$(for x in $X; do printf "%d," $((1000 + x)); done | sed 's/,$//')
V2=$(for x in $X; do printf "%d," $((2000 + x)); done | sed 's/,$//')
timeout 100 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 60 --variants $V4,$V2 --sweeps 2 --reps 5 > $OUT/t_rpb60_2m.jsonl 2> $OUT/t_rpb60_2m.err
timeout 100 python tools/bench_decode.py --lib $LIB --records 4000000 --rpb 60 --variants $V4,$V2 --sweeps 2 --reps 5 > $OUT/t_rpb60_4m.jsonl 2> $OUT/t_rpb60_4m.err
timeout 100 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 500 --variants $V2,1000 --sweeps 2 --reps 5 > $OUT/t_rpb500_2m.jsonl 2> $OUT/t_rpb500_2m.err
timeout 100 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 8 --variants $V4 --sweeps 2 --reps 5 > $OUT/t_rpb8_2m.jsonl 2> $OUT/t_rpb8_2m.err
timeout 100 python tools/bench_decode.py --lib $LIB --records 2000000 --rpb 30,120 --variants 1000,1259,2283,2000,2259,3283 --sweeps 2 --reps 5 > $OUT/t_rpb30_120_2m.jsonl 2> $OUT/t_rpb30_120.err
wc -l $OUT/*.jsonl
tail -n 3 $OUT/*.err
