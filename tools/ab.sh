#!/bin/bash
# A/B of library variants on one box (tools/variant.sh builds them): alternating runs of tools/perf.py.
#   tools/ab.sh "<perf.py arguments>" <tag> [<tag> ...]      tag "shipped" = swarm_amd/lib/libswarm_hip.so
# prints one line per run: tag, kernel ms (median repetition), min..max
root=$(cd "$(dirname "$0")/.." && pwd)
args=$1; shift
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = shipped ]; then lib=$root/swarm_amd/lib/libswarm_hip.so; else lib=$root/swarm_amd/lib/variants/libswarm_hip_$tag.so; fi
    SWARM_HIP_LIB=$lib timeout -k 10 120 python3 $root/tools/perf.py $args --warm --no-host-to-host --reps 5 > /tmp/ab_$tag.json || exit 1
    python3 -c "import json;d=json.load(open('/tmp/ab_$tag.json'));print('$tag', round(d['kernel_ms_all_launches'],4), d['kernel_ms_all_launches_min_max'], d['accepted'])"
  done
done
