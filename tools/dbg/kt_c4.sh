#!/bin/bash
# GPU box only: kernel-trace summary of the c4 step under the current environment -> gpurun_out/<tag>_c4_kernel_stats.csv
set -e -o pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $root/bench.py --config c4 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $out/${tag}_c4_profiled.json 2> $out/${tag}_c4_kt.err
python3 $root/tools/prof_summary.py /tmp/prof_kt/kt_results.db $out/${tag}_c4_kernel_stats.csv > $out/${tag}_c4_top_kernels.txt
rm -rf /tmp/prof_kt
