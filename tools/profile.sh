#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun, from the repo root):
#   tools/profile.sh <tag> <program args after "python3">
# e.g. tools/profile.sh hermite_n3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
# One kernel-trace/stats pass, then one --pmc pass per counter block (rocprofv3 does not split passes).
# Output: gpurun_out/prof_<tag>/{stats,sq,sq2,fetch,write}/... ; tools/pmc_summary.py condenses it.
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$root
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/$1" "${@:2}" > "$out/stats.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- python3 "$root/$1" "${@:2}" > "$out/sq.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$out/sq2" -- python3 "$root/$1" "${@:2}" > "$out/sq2.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/$1" "${@:2}" > "$out/fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 "$root/$1" "${@:2}" > "$out/write.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py "$out" > "$out/summary.json"
cat "$out/summary.json"
