#!/bin/bash
# GPU box only: the other configs' bench lines under the current build -> gpurun_out/<tag>_bench_<cfg>.json
tag=$1
for c in c1 c3 c5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || exit 1; done
timeout -k 10 300 python bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/${tag}_bench_fp32.json 2> gpurun_out/${tag}_bench_fp32.err || exit 1
timeout -k 10 400 python bench.py --config c4 > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err || exit 1
