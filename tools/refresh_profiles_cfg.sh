#!/bin/bash
# GPU box only: kernel trace + FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --config <cfg>` (c3, c5, ft ...),
# summaries into gpurun_out/<tag>_*.   usage (through gpurun): bash tools/refresh_profiles_cfg.sh r03x c3
set -e -o pipefail
tag=${1:-rXX}; cfg=${2:-c3}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps="--steps 40 --warmup 5"; if [ "$cfg" = "ft" ]; then steps="--steps 4 --warmup 2"; fi
# c4: the counter passes launch the step eagerly (same kernels, same operands): replaying the captured training step under
# `rocprofv3 --pmc` dies with SIGSEGV on a device address inside hipGraphLaunch (r03n_c4_pf.err; the kernel trace of the
# same graph is fine, and so are the nav-step graphs under --pmc)
pmc_extra=""; if [ "$cfg" = "c4" ]; then pmc_extra="--eager"; fi
rm -rf /tmp/prof_kt /tmp/prof_pf /tmp/prof_pw
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $root/bench.py --config $cfg $steps --no-cpu-baseline --no-roofline > $out/${tag}_bench_${cfg}_profiled.json 2> $out/${tag}_${cfg}_kt.err
python3 $root/tools/prof_summary.py /tmp/prof_kt/kt_results.db $out/${tag}_bench_${cfg}_kernel_stats.csv --seq > $out/${tag}_bench_${cfg}_step_sequence.txt
rm -rf /tmp/prof_kt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_pf -o pf -- python3 $root/bench.py --config $cfg $steps $pmc_extra --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_${cfg}_pf.err
python3 $root/tools/pmc_summary.py /tmp/prof_pf/pf_results.db $out/${tag}_${cfg}_pmc_fetch.csv > /dev/null
rm -rf /tmp/prof_pf
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_pw -o pw -- python3 $root/bench.py --config $cfg $steps $pmc_extra --no-cpu-baseline --no-roofline > /dev/null 2> $out/${tag}_${cfg}_pw.err
python3 $root/tools/pmc_summary.py /tmp/prof_pw/pw_results.db $out/${tag}_${cfg}_pmc_write.csv > /dev/null
rm -rf /tmp/prof_pw
echo refreshed $tag $cfg
