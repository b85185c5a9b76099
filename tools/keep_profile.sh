#!/bin/bash
# Copy the judged parts of a tools/profile.sh run from the scratch directory into the tracked profiles/<round>/:
#   tools/keep_profile.sh <tag> <round>      e.g. tools/keep_profile.sh hermite_n3 r2
# -> profiles/<round>/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_rocprof_summary.json
#    (tools/pmc_summary.py: kernels, per-launch min/median/max, counters, HBM bytes), <tag>_pmc.json (counters only)
set -e
tag=$1; rnd=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gpurun_out/prof_$tag
dst=$root/profiles/$rnd
mkdir -p "$dst"
cp "$(ls "$src"/stats/*/*kernel_stats.csv | head -1)" "$dst/${tag}_kernel_stats.csv"
cp "$src/summary.json" "$dst/${tag}_rocprof_summary.json"
python3 -c "import json,sys; d=json.load(open('$src/summary.json')); json.dump(d['counters'], open('$dst/${tag}_pmc.json','w'), indent=1)"
ls -la "$dst" | grep "$tag"
