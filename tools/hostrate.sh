#!/bin/bash
# host-side rate of the per-message entry (what a Rust MetricHandler shim pays per polled message)
cd $GRAFT_REPO_ROOT
N=20000000
for mode in "kta.per_message=1" "kta.batch=1048576"; do
  for c in "" "-c"; do
    t0=$(date +%s.%N)
    kafka_topic_analyzer_amd/kta-analyzer -t c2 -b "synthetic://c2?records=$N" $c --librdkafka $mode > /tmp/out.txt 2>/tmp/err.txt
    rc=$?
    t1=$(date +%s.%N)
    python3 -c "print('rc=$rc  %.2f s  %.1f M records/s   [$mode $c]' % ($t1-$t0, $N/($t1-$t0)/1e6))"
    grep -E "Alive keys" /tmp/out.txt
  done
done
