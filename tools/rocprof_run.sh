#!/bin/bash
# rocprofv3 recipe (run on the GPU box through gpurun):  tools/rocprof_run.sh <tag> [stats|pmc|all] -- <command...>
# Writes text summaries to gpurun_out/prof_<tag>/{stats.txt,pmc.txt}; counters run in their own passes
# (--pmc is never combined with anything but --kernel-trace).
TAG=$1; MODE=$2; shift 3
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp
if [ "$MODE" = stats ] || [ "$MODE" = all ]; then
  rocprofv3 --kernel-trace --stats -d $RAW/stats -o r -- "$@" > $OUT/stats_cmd.out 2> $OUT/stats_cmd.err
  python $ROOT/tools/summarize_rocprof.py stats $(find $RAW/stats -name '*.db' | head -1) > $OUT/stats.txt 2>&1
fi
if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $RAW/pmc_$c -o r -- "$@" > $OUT/pmc_${c}_cmd.out 2> $OUT/pmc_${c}_cmd.err
  done
  python $ROOT/tools/summarize_rocprof.py pmc $(find $RAW/pmc_* -name '*.db') > $OUT/pmc.txt 2>&1
fi
