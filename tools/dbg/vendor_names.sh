#!/bin/bash
# GPU box only: which vendor-library kernels run the wide nav-step GEMM shapes (names carry the macro tile)
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/vendor_names.txt
for s in "l.ff1" "l.qkv" "l.cq" "l.ff2" "g.ff1" "square"; do
  rm -rf /tmp/vn
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/vn -o vn -- python3 $root/tools/bench_gemm.py --blas --cold --only "$s" >> $out/vendor_names.txt 2> $out/vendor_names.err
  python3 $root/tools/prof_summary.py /tmp/vn/vn_results.db /tmp/vn/stats.csv > /dev/null
  head -6 /tmp/vn/stats.csv | cut -c1-260 >> $out/vendor_names.txt
done
