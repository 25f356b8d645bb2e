#!/bin/bash
# A/B of build switches of the alive pass (kta_alive.hip) with tools/ubench_alive.hip, many variants per gpurun call
# (a call costs about half a GPU-minute before the command starts).
#
#   build container:   tools/ab_alive.sh build base: cand:"-DSOME_SWITCH_OF_THE_CANDIDATE"      (a candidate's switch lives in the tree only while it is
#                      being measured: round 4's were deleted with their timings kept in profiles/r04_ab_alive.txt)
#                      -> tools/ubench_alive_ab_<tag>, one binary per "tag:flags" (timing builds: no phase counters)
#   GPU box:           gpurun --timeout 120 -- 'AB_FULL=tools/ubench_alive_ab_base bash tools/ab_alive.sh run > gpurun_out/ab.txt 2>&1'
#                      -> per binary: best of 4 repetitions (whole pass and per kernel) on the headline workload, and whether
#                         the alive count is the known one; the binaries named in AB_FULL also run the other workloads
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
    # every binary on the headline workload (bit set state); the full matrix only for the binaries named in $AB_FULL
    # (default: the first one).  Binaries built with switches that time a part of the work leave the counts wrong
    # by design.
    one() {     # binary, log2 n, distinct, expected count, state
        out=$(timeout 60 "$1" "$2" "$3" "$5" 2>&1)
        best=$(echo "$out" | grep "^rep" | awk '{ if (min == "" || $3 < min) min = $3 } END { print min }')
        part=$(echo "$out" | grep "^kernels" | awk '{ if (min == "" || $4 < min) min = $4 } END { print min }')
        appl=$(echo "$out" | grep "^kernels" | awk '{ if (min == "" || $6 < min) min = $6 } END { print min }')
        fall=$(echo "$out" | grep "^kernels" | awk '{ if (max == "" || $8 > max) max = $8 } END { print max }')
        bad=$(echo "$out" | grep "^rep" | grep -vc "alive=$4 ")
        printf "state %s  n=2^%s distinct=%-10s %-20s best %8s ms  partition %7s  apply %7s  fallback(max) %7s  %s\n" "$5" "$2" "$3" \
            "${1#tools/ubench_alive_ab_}" "$best" "$part" "$appl" "$fall" "$([ "$bad" = 0 ] && [ -n "$best" ] && echo "count ok" || echo "COUNT WRONG OR NO OUTPUT")"
        echo "$out" | grep -i "error\|fault\|abort" | head -3
    }
    for bin in tools/ubench_alive_ab_*; do one "$bin" 28 10000000 8988944 0; done
    full=${AB_FULL:-$(ls tools/ubench_alive_ab_* | head -1)}
    for bin in $full; do
        one "$bin" 26 10000000 8978440 0
        one "$bin" 26 100000000 43746097 0
        one "$bin" 28 100000000 0 0
        one "$bin" 26 10000000 8978440 1
        one "$bin" 26 100000000 43746097 1
    done
    exit 0
fi
sed -n 2,16p "$0"
