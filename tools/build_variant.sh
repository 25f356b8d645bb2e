#!/bin/bash
# A second build of the library beside the product's: kafka_topic_analyzer_amd/libkta_hip.so.<tag>, from the sources
# as they lie in the tree plus extra compiler flags (build switches of an experiment).  git-ignored; travels to the
# GPU box with the snapshot.  tools/bench_decode.py --lib <file>[,<file>] times several builds in one gpurun call.
#   tools/build_variant.sh <tag> [-DSWITCH=1 ...]
set -eu
cd "$(dirname "$0")/.."
tag=$1; shift
csrc=kafka_topic_analyzer_amd/csrc
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
common="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I include -I $csrc"
for s in kta_kernels kta_alive kta_api kta_comm kta_synth kta_kafka; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $common "$@" -c $csrc/$s.hip -o $obj/$s.o &
done
for s in metric report kafka_encode; do
    /opt/rocm/bin/hipcc $common "$@" -c $csrc/host/$s.cpp -o $obj/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kafka_topic_analyzer_amd/libkta_hip.so.$tag $obj/*.o -ldl -Wl,-rpath,/opt/rocm/lib
echo "built kafka_topic_analyzer_amd/libkta_hip.so.$tag ($*)"
