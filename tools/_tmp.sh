#!/bin/bash
# temporary: full GPU suite, bench, nbod 17..32 record
set -o pipefail
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests_nbod32.txt 2>&1 || { tail -30 gpurun_out/gpu_tests_nbod32.txt; exit 1; }
tail -3 gpurun_out/gpu_tests_nbod32.txt
timeout -k 10 300 python bench.py > gpurun_out/bench_nbod32.json 2> gpurun_out/bench_nbod32.err || { tail gpurun_out/bench_nbod32.err; exit 1; }
out=gpurun_out/nbod_17_32_perf.txt
: > $out
fmt='import json,sys;d=json.loads(sys.stdin.read());print(d["integrator"], "nbod", d["nbod"], d["nsys"], "ms", round(d["ms"],2), d["ms_min_max"], "steps/s %.3g" % d["accepted_steps_per_s"], "frac", round(d["frac_of_fp64_peak"],3))'
for n in 17 21 22 24 32; do
  timeout -k 10 120 python tools/perf.py c2 --nbod $n --nsys 65536 --T 1 --reps 3 | python3 -c "$fmt" >> $out || exit 1
done
for n in 17 24 32; do
  timeout -k 10 120 python tools/perf.py c2 --nbod $n --nsys 2048 --T 1 --reps 3 | python3 -c "$fmt" >> $out || exit 1
  timeout -k 10 120 python tools/perf.py c4 --nbod $n --nsys 16384 --reps 3 | python3 -c "$fmt" >> $out || exit 1
  timeout -k 10 120 python tools/perf.py c3_nomon --nbod $n --nsys 16384 --reps 3 | python3 -c "$fmt" >> $out || exit 1
done
cat $out
