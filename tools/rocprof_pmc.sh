#!/bin/bash
# rocprofv3 counter passes (run on the GPU box through gpurun):
#   tools/rocprof_pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>
# One rocprofv3 run per quoted counter list (--pmc with --kernel-trace only); summary -> gpurun_out/prof_<tag>/pmc.txt
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp
i=0
for p in "${PASSES[@]}"; do
  rocprofv3 --pmc $p --kernel-trace -d $RAW/pmc_$i -o r -- "$@" > $OUT/pmc_${i}_cmd.out 2> $OUT/pmc_${i}_cmd.err
  i=$((i+1))
done
python $ROOT/tools/summarize_rocprof.py pmc $(find $RAW/pmc_* -name '*.db') > $OUT/pmc.txt 2>&1
