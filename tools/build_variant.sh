#!/bin/bash
# Builds a variant of libdiffgraph_hip.so with extra compiler flags for spmm.hip into scratch/libdg_<tag>.so
# (for A/B timing on the GPU box: copy it over diffgraph_amd/libdiffgraph_hip.so in that run only).
#   tools/build_variant.sh <tag> [flags...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
python -m diffgraph_amd.build >/dev/null
C=diffgraph_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $C/spmm.hip -o scratch/spmm_$tag.o
objs=""
for s in graph diffusion wgrad losshead eval sampling spgemm optim; do objs="$objs $C/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdg_$tag.so scratch/spmm_$tag.o $objs
echo scratch/libdg_$tag.so
