#!/bin/bash
# GPU box only: kernel-trace timeline of one c2 step under the environment given as arguments (VAR=.. ...)
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python3 $root/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > $root/gpurun_out/timeline_$tag.json 2> $root/gpurun_out/timeline_$tag.err
python3 $root/tools/dbg/timeline.py /tmp/tl/tl_results.db > $root/gpurun_out/timeline_$tag.txt
