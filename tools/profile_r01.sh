#!/bin/bash
# Round-1 profile recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats,
# and HBM traffic counters in their own passes.  Summaries are copied to profiles/ by hand.
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/prof
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 > $OUT/bench_r01.json 2> $OUT/bench_r01.err
tail -c 3000 $OUT/bench_r01.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof/stats -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof/stats_bench.json 2> $OUT/prof/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof/pmc_fetch -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof/pmc_fetch_bench.json 2> $OUT/prof/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof/pmc_write -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof/pmc_write_bench.json 2> $OUT/prof/pmc_write.err
find $OUT/prof -type f | head -50
du -sh $OUT/prof
