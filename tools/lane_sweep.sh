# lane-per-system (1) vs lane-per-body (2) kernel time over nsys; run on the GPU box from the repo root
#   WORKLOAD=c2 (Hermite, default), c4 (MVS), irk2, c3 (rkck_adaptive), euler, midpoint, hermite_prop or verlet; NBODS, NSYS override the grids
W=${WORKLOAD:-c2}; T=0.5; [ "$W" = c4 ] && T=5.0; [ "$W" = irk2 ] && T=2.0; [ "$W" = c3 ] && T=2.0; [ "$W" = c3_nomon ] && T=2.0; [ -n "$TEND" ] && T=$TEND
for n in ${NBODS:-3 4 5 6 8 10}; do for nsys in ${NSYS:-64 2048 8192 16384 24576 32768 49152 65536}; do for m in 1 2; do
timeout -k 10 100 python tools/perf.py $W --T $T --nbod $n --nsys $nsys --lane-mode $m --reps 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['nbod'], d['nsys'], $m, round(d['ms'],3))" || exit 1
done; done; done
