#!/bin/bash
# GPU box only: kernel trace of `python3 bench.py --config <cfg>` (environment passes through), one step's sequence and
# per-kernel stats into gpurun_out/<tag>_*.   usage: bash tools/dbg/kt_c2.sh <tag> [cfg]
set -e -o pipefail
tag=${1:-rXX}; cfg=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$tag -o kt -- python3 $root/bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > $out/${tag}_bench_${cfg}_profiled.json 2> $out/${tag}_kt.err
python3 $root/tools/prof_summary.py /tmp/prof_kt_$tag/kt_results.db $out/${tag}_bench_${cfg}_kernel_stats.csv --seq > $out/${tag}_bench_${cfg}_step_sequence.txt
rm -rf /tmp/prof_kt_$tag
echo traced $tag
