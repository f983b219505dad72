#!/bin/bash
# GPU box only: A/B of the c2 step under environment variants, same box, alternating; prints ms_per_step per variant
out=gpurun_out/ab_c2.txt; : > $out
run() { name=$1; shift; r=$(env "$@" timeout -k 10 150 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "$name $r" | tee -a $out; }
for rep in 1 2; do
for v in "$@"; do run "$v" $v; done
done
