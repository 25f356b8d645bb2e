#!/bin/bash
# A/B of build switches of the alive pass (kta_alive.hip) with tools/ubench_alive.hip, many variants per gpurun call
# (a call costs about half a GPU-minute before the command starts).
#
#   build container:   tools/ab_alive.sh build base: one:-DKTA_APPLY_SITES=1 stage0:-DKTA_PART_STAGE=0
#                      -> tools/ubench_alive_ab_<tag>, one binary per "tag:flags" (timing builds: no phase counters)
#   GPU box:           gpurun --timeout 120 -- 'bash tools/ab_alive.sh run > gpurun_out/ab.txt 2>&1'
#                      -> per binary and workload: best of 4 repetitions, and whether the alive count is the known one
#
# Workloads (ubench arguments) and their alive counts, which do not depend on the state or on any switch:
#   28 10000000   compacted topic at the size of bench.py's alive pass       8988944
#   26 10000000   the same at 2^26                                           8978440
#   26 100000000  mostly unique keys (restart in careful mode, instalments)  43746097
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
    shift
    rm -f tools/ubench_alive_ab_*
    for spec in "$@"; do
        tag=${spec%%:*}
        flags=${spec#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKTA_NO_PHASES $flags -I include \
            -I kafka_topic_analyzer_amd/csrc tools/ubench_alive.hip -o tools/ubench_alive_ab_$tag 2>/dev/null || { echo "build failed: $spec"; exit 1; }
        echo "built tools/ubench_alive_ab_$tag  ($flags)"
    done
    exit 0
fi
if [ "${1:-}" = run ]; then
    for state in 0 1; do
        for w in "28 10000000 8988944" "26 10000000 8978440" "26 100000000 43746097"; do
            set -- $w
            [ "$state" = 1 ] && [ "$1" = 28 ] && continue       # (table state: the 2^26 workloads say enough)
            for bin in tools/ubench_alive_ab_*; do
                out=$(timeout 30 "$bin" "$1" "$2" "$state" 2>&1 | grep "^rep")
                best=$(echo "$out" | awk '{ if (min == "" || $3 < min) min = $3 } END { print min }')
                bad=$(echo "$out" | grep -vc "alive=$3 ")
                printf "state %s  n=2^%s distinct=%-10s %-28s best %8s ms  %s\n" "$state" "$1" "$2" "${bin#tools/ubench_alive_ab_}" "$best" \
                    "$([ "$bad" = 0 ] && [ -n "$best" ] && echo "count ok" || echo "COUNT WRONG OR NO OUTPUT")"
            done
        done
    done
    exit 0
fi
sed -n 2,16p "$0"
