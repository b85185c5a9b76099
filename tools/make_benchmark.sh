#!/bin/bash
# The reference's `make benchmark` target (cmake/custom.cmake:7-40) with this repo's CLI:
#   swarm benchmark --range nsys=1000,2000,4000,8000,16000 nbod=N -c test/benchmark/Hermite.cfg   (N = 3, 4, 5)
#   swarm benchmark --range systems_ber_block=1,2,4,8,16,32 nsys=16000 nbod=3 -c test/benchmark/Hermite.cfg
# Run on the GPU box from the repo root: tools/make_benchmark.sh <output directory> [extra key=value ...]
set -e -o pipefail
out=${1:?output directory}; shift
mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
cfg=$root/tests/golden/benchmark/Hermite.cfg
for n in 3 4 5; do
    timeout -k 10 300 "$root/swarm_amd/bin/swarm" benchmark --range nsys=1000,2000,4000,8000,16000 nbod=$n -c "$cfg" "$@" \
        2> "$out/benchmark_Hermite_$n.log" > "$out/benchmark_Hermite_$n.csv"
    echo "benchmark_Hermite_$n done"
done
timeout -k 10 300 "$root/swarm_amd/bin/swarm" benchmark --range systems_ber_block=1,2,4,8,16,32 nsys=16000 nbod=3 -c "$cfg" "$@" \
    2> "$out/benchmark_block_Hermite.log" > "$out/benchmark_block_Hermite.csv"
echo "benchmark_block_Hermite done"
