#!/bin/bash
# GPU box only: kernel trace + PMC passes of `python3 bench.py --config c4` (the pre-training step), summaries into gpurun_out/<tag>_*.
# usage (through gpurun): bash tools/refresh_profiles_c4.sh r02k
set -e -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_pf /tmp/prof_pw
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $root/bench.py --config c4 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $out/${tag}_bench_c4_profiled.json 2> $out/${tag}_c4_kt.err
python3 $root/tools/prof_summary.py /tmp/prof_kt/kt_results.db $out/${tag}_bench_c4_kernel_stats.csv > $out/${tag}_bench_c4_top_kernels.txt
rm -rf /tmp/prof_kt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_pf -o pf -- python3 $root/bench.py --config c4 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_c4_pf.err
python3 $root/tools/pmc_summary.py /tmp/prof_pf/pf_results.db $out/${tag}_c4_pmc_fetch.csv > /dev/null
rm -rf /tmp/prof_pf
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_pw -o pw -- python3 $root/bench.py --config c4 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_c4_pw.err
python3 $root/tools/pmc_summary.py /tmp/prof_pw/pw_results.db $out/${tag}_c4_pmc_write.csv > /dev/null
rm -rf /tmp/prof_pw
echo refreshed $tag c4
