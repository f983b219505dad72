set -o pipefail
tag=r03n
bash tools/refresh_profiles_cfg.sh $tag c4 || exit 1
bash tools/dbg/bench_all.sh $tag || exit 1
timeout -k 10 400 python bench.py --config ft > gpurun_out/${tag}_bench_ft.json 2> gpurun_out/${tag}_bench_ft.err || exit 1
for c in c2 c4 ft; do
  st="--steps 30 --warmup 5"; [ $c = ft ] && st="--steps 4 --warmup 2"
  DSRG_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --config $c --gpus 2 $st --no-cpu-baseline --no-roofline > gpurun_out/${tag}_bench_${c}_2ranks_gloo_one_gpu.json 2> gpurun_out/${tag}_bench_${c}_2ranks.err || exit 1
done
echo evidence part 2 done
