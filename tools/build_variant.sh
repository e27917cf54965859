#!/bin/bash
# study builds: the library with extra preprocessor flags into an alternate file for same-box A/B runs (IVIT_LIB=<file> python bench.py ...)
# usage: tools/build_variant.sh tools/libivit_x.so -DIVIT_SOMETHING=1 ...
set -e
out=$1; shift
tmp=$(mktemp -d)
for f in engine kernels_gemm kernels_attn kernels_misc kernels_mlp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c interactive_vit_amd/csrc/$f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $tmp/*.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libivit.so
rm -rf $tmp
