#!/bin/bash
# Build a variant of libdsrg_hip.so with extra compiler flags for same-box A/B runs (select it with DSRG_HIP_LIB=<path>).
# usage: tools/build_variant.sh <name> [-DFLAG ...]   ->  dsrg_amd/csrc/variants/libdsrg_hip_<name>.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/dsrg_amd/csrc; out=$src/variants; mkdir -p $out/obj_$name
for f in gemm rowops attention train train_rows graphmap optim; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$root/include "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdsrg_hip_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo $out/libdsrg_hip_$name.so
