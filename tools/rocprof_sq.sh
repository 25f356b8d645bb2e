#!/bin/bash
# SQ counters of a command's kernels (one --pmc pass of 8 SQ slots with --kernel-trace only):
#   tools/rocprof_sq.sh <tag> -- <command...>      -> gpurun_out/prof_<tag>/sq.txt
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count issued
# wave instructions.  insts / waves = dynamic instructions per wave; WAIT_ANY = parked on s_waitcnt / barrier.
TAG=$1; shift 2
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_sq_$TAG
mkdir -p $OUT $RAW
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --kernel-trace -d $RAW -o r -- "$@" > $OUT/sq_cmd.out 2> $OUT/sq_cmd.err
python $ROOT/tools/summarize_rocprof.py pmc $(find $RAW -name '*.db') > $OUT/sq.txt 2>&1
