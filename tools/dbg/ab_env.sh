#!/bin/bash
# GPU box only: same-box A/B of bench.py under environment variants (one quoted "VAR=.. VAR=.." string per variant; "-" = none)
# usage: bash tools/dbg/ab_env.sh <config> <reps> "<env string>" ...
cfg=$1; reps=$2; shift; shift
out=gpurun_out/ab_env_$cfg.txt; : > $out
for rep in $(seq $reps); do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  r=$(env $e timeout -k 10 200 python bench.py --config $cfg $DSRG_AB_EXTRA --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "[$v] $r" | tee -a $out
done
done
