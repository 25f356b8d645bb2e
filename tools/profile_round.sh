#!/bin/bash
# (the counter passes run with --no-alive-extras: every kernel at ONE launch size, so a row's average is that launch)
# Per-round profile recipe (tools/profile_round.sh rNN) (run on the GPU box through gpurun): the bench line, rocprofv3 kernel stats of the same
# command, and the HBM traffic counters in their own passes.  tools/make_traffic.py turns pmc.txt into
# profiles/traffic.json; the text summaries are copied to profiles/ by hand.
TAG=${1:-r06}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${TAG}
RAW=/tmp/prof_${TAG}
mkdir -p $OUT $RAW
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $RAW/stats -o r -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hostfed --no-alive-extras > $OUT/bench_stats_run.json 2> $OUT/stats.err
python $ROOT/tools/summarize_rocprof.py stats $(find $RAW/stats -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $RAW/pmc_$c -o r -- python $ROOT/bench.py --steps 5 --warmup 1 --preroll 5 --no-cpu-baseline --no-hostfed --no-alive-extras > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
done
python $ROOT/tools/summarize_rocprof.py pmc $(find $RAW/pmc_* -name '*.db') > $OUT/pmc_hbm.txt 2>&1
# the default bench line LAST, with roofline.traffic replayed from THESE counter passes (the same file is regenerated from
# pmc_hbm.txt in the repo afterwards: tools/make_traffic.py records the sources' hashes, which are the same there)
cd $ROOT
python tools/make_traffic.py $OUT/pmc_hbm.txt > profiles/traffic.json 2> $OUT/make_traffic.err
timeout 500 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.err
# the multi-GPU modes with ONE rank and the collectives forced (a 1-GPU box): config 5 (both handlers + the whole
# exchange) and config 4 as stated (strong scaling); "forced_collectives": true, not headline numbers
cd $ROOT
export KTA_BENCH_FORCE_COLLECTIVES=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --preroll 5 --config c5 > $OUT/bench_c5_forced.json 2> $OUT/bench_c5_forced.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --scaling strong --no-alive --no-decode --no-hostfed --no-cpu-baseline > $OUT/bench_c4_strong_forced.json 2> $OUT/bench_c4_strong_forced.err
ls -la $OUT
