#!/bin/bash
# Round-3 profile recipe (run on the GPU box through gpurun): the bench line, rocprofv3 kernel stats of the same
# command, and the HBM traffic counters in their own passes.  tools/make_traffic.py turns pmc.txt into
# profiles/traffic.json; the text summaries are copied to profiles/ by hand.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r03
RAW=/tmp/prof_r03
mkdir -p $OUT $RAW
cd $ROOT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $RAW/stats -o r -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_stats_run.json 2> $OUT/stats.err
python $ROOT/tools/summarize_rocprof.py stats $(find $RAW/stats -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $RAW/pmc_$c -o r -- python $ROOT/bench.py --steps 5 --warmup 1 --preroll 5 --no-cpu-baseline > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
done
python $ROOT/tools/summarize_rocprof.py pmc $(find $RAW/pmc_* -name '*.db') > $OUT/pmc_hbm.txt 2>&1
ls -la $OUT
