#!/bin/bash
# First GPU call of a round whose tree holds a decode kernel the builder has not run on the GPU yet:
# its parity gate, its timing per batch size and geometry, then the round's evidence (tools/profile_round.sh writes
# gpurun_out/prof_rNN; copy the summaries to profiles/rNN_* and rerun tools/make_traffic.py on the SAME tree).
#   gpurun --timeout 2400 -- bash tools/first_call.sh
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/first_call
mkdir -p $OUT
cd $ROOT
# (the GPU tests that run the decode kernel: its own file, the CLI's segment:// source, the consumer loop's double, the demo)
[ -n "$SKIP_PYTEST" ] || timeout 900 python -m pytest tests/test_kafka_decode.py tests/test_report_cli.py tests/test_rdkafka_loop.py tests/test_reference_demo.py -x -q -m gpu > $OUT/pytest_decode.txt 2>&1
tail -3 $OUT/pytest_decode.txt
# automatic choice (0) and the three geometries that are left (round 5: profiles/r05_decode_geometries.jsonl has the ones that lost):
# 10 = <4, 3 KiB, 16>, 11 = <2, 8 KiB, 32>, 2 = <1, 8 KiB, 256>; ~2 / ~16 / ~134 KiB batches and ~10 KiB RECORDS in 64 KiB batches
timeout 300 python tools/bench_decode.py --rpb 8,30,60,90,120,250,500 --variants 0,10,11,2 --reps 5 > $OUT/bench_decode.txt 2>&1
timeout 120 python tools/bench_decode.py --records 4000000 --rpb 60,120 --variants 10,11 --reps 5 >> $OUT/bench_decode.txt 2>&1
timeout 120 python tools/bench_decode.py --records 100000 --rpb 6,24 --val-mean 10240 --variants 0,10,11,2 --reps 5 >> $OUT/bench_decode.txt 2>&1
tail -50 $OUT/bench_decode.txt
bash tools/profile_round.sh r05 > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
