#!/bin/bash
# GPU box only: the round's evidence set under the current build -> gpurun_out/<tag>_*.
#   c2: kernel trace + 4 PMC passes, bench line;  c3 / c5 / c4: kernel trace + FETCH / WRITE passes;  bench lines of every config;
#   two-rank rehearsals (two ranks sharing the one GPU over gloo) of c2, c4 and ft.
# usage (through gpurun): bash tools/final_evidence.sh r03n
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
bash tools/refresh_profiles.sh $tag || exit 1
for c in c3 c5 c4; do bash tools/refresh_profiles_cfg.sh $tag $c || exit 1; done
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || exit 1
bash tools/dbg/bench_all.sh $tag || exit 1
timeout -k 10 400 python bench.py --config ft > gpurun_out/${tag}_bench_ft.json 2> gpurun_out/${tag}_bench_ft.err || exit 1
for c in c2 c4 ft; do
  st="--steps 30 --warmup 5"; [ $c = ft ] && st="--steps 4 --warmup 2"
  DSRG_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --config $c --gpus 2 $st --no-cpu-baseline --no-roofline > gpurun_out/${tag}_bench_${c}_2ranks_gloo_one_gpu.json 2> gpurun_out/${tag}_bench_${c}_2ranks.err || exit 1
done
echo evidence $tag done
