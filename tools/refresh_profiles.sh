#!/bin/bash
# GPU box only: kernel trace + PMC passes of `python3 bench.py` (the default command), summaries into gpurun_out/<tag>_*.
# usage (through gpurun): bash tools/refresh_profiles.sh r01d
set -e -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_pf /tmp/prof_pw /tmp/prof_pt /tmp/prof_pm
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $root/bench.py --no-cpu-baseline > $out/${tag}_bench_c2_profiled.json 2> $out/${tag}_kt.err
python3 $root/tools/prof_summary.py /tmp/prof_kt/kt_results.db $out/${tag}_bench_c2_kernel_stats.csv --seq > $out/${tag}_bench_c2_step_sequence.txt
rm -rf /tmp/prof_kt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_pf -o pf -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pf.err
python3 $root/tools/pmc_summary.py /tmp/prof_pf/pf_results.db $out/${tag}_pmc_fetch.csv > /dev/null
rm -rf /tmp/prof_pf
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_pw -o pw -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pw.err
python3 $root/tools/pmc_summary.py /tmp/prof_pw/pw_results.db $out/${tag}_pmc_write.csv > /dev/null
rm -rf /tmp/prof_pw
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/prof_pt -o pt -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pt.err
python3 $root/tools/pmc_summary.py /tmp/prof_pt/pt_results.db $out/${tag}_pmc_tcc.csv > /dev/null
rm -rf /tmp/prof_pt
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d /tmp/prof_pm -o pm -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_pm.err
python3 $root/tools/pmc_summary.py /tmp/prof_pm/pm_results.db $out/${tag}_pmc_mfma.csv > /dev/null
rm -rf /tmp/prof_pm
echo refreshed $tag
