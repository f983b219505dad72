#!/bin/bash
# GPU box only: A/B of the c2 step under environment variants, same box, alternating; prints ms_per_step per variant
out=gpurun_out/ab_c2.txt; : > $out
run() { name=$1; shift; r=$(env "$@" timeout -k 10 150 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "$name $r" | tee -a $out; }
for rep in 1 2; do
run A_default X=1
run B_grouped DSRG_SPLIT_BRANCHES=0
run C_grouped_nofuse DSRG_SPLIT_BRANCHES=0 DSRG_FUSE_LN=0
run D_grouped_nofuse_128x128x2 DSRG_SPLIT_BRANCHES=0 DSRG_FUSE_LN=0 DSRG_GEMM_WIDE=300:128x128x2
run E_grouped_nofuse_128x128x3 DSRG_SPLIT_BRANCHES=0 DSRG_FUSE_LN=0 DSRG_GEMM_WIDE=300:128x128x3
run F_grouped_nofuse_256x128x3 DSRG_SPLIT_BRANCHES=0 DSRG_FUSE_LN=0 DSRG_GEMM_WIDE=300:256x128x3
run G_split_nofuse DSRG_FUSE_LN=0
run H_split_nofuse_128x128x2 DSRG_FUSE_LN=0 DSRG_GEMM_WIDE=300:128x128x2
done
