#!/bin/bash
# The reference's saturation sweep (scripts/saturation/by_integrators.sh + ranges.py) with this repo's CLI:
#   swarm benchmark -c samples/Hermite.cfg --range nsys=32,45.25,...,32768 nbod=N -d 2.0    for N in 3 4 6 8 10
# Run on the GPU box from the repo root: tools/saturation.sh <output directory>
set -e -o pipefail
out=${1:?output directory}
mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
nsys=$(python3 -c "
import math
p, x = 32, []
for i in range(1, 22):
    x.append(p); p = math.sqrt(2.0) * p
print(','.join(map(str, x)))")
for n in 3 4 6 8 10; do
    timeout -k 10 300 "$root/swarm_amd/bin/swarm" benchmark -c "$root/tests/golden/samples/Hermite.cfg" --range nsys=$nsys \
        nsys=2000 nbod=$n -d 2.0 2> "$out/hermite_$n-log.txt" > "$out/hermite_$n.csv"
    echo "nbod=$n done: $(wc -l < "$out/hermite_$n.csv") lines"
done
