#!/bin/bash
# GPU box only: same-box A/B of bench.py between library builds (DSRG_HIP_LIB), alternating.
# usage: bash tools/dbg/ab_lib.sh <config> <lib-or-"-"> ...   ("-" = the in-tree library)
cfg=$1; shift
out=gpurun_out/ab_lib_$cfg.txt; : > $out
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="DSRG_HIP_LIB=$v"; fi
  r=$(env $e timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$v $r" | tee -a $out
done
done
