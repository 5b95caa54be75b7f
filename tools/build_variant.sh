#!/bin/bash
# Builds a variant of libdiffgraph_hip.so with extra compiler flags for spmm.hip and graph.hip (or the sources named in
# VARIANT_SOURCES) into
# scratch/libdg_<tag>.so (for A/B timing on the GPU box: copy it over diffgraph_amd/libdiffgraph_hip.so in that run only, and
# run with DG_ALLOW_STALE_LIB=1: a variant carries no source digest, so diffgraph_amd._lib would refuse it).
#   tools/build_variant.sh <tag> [flags...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
python -m diffgraph_amd.build >/dev/null
mkdir -p scratch
C=diffgraph_amd/csrc
for s in ${VARIANT_SOURCES:-spmm graph}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $C/$s.hip -o scratch/${s}_$tag.o
done
objs=""
for s in graph spmm fused diffusion wgrad losshead eval sampling spgemm optim linear comm; do
  if [ -f scratch/${s}_$tag.o ]; then objs="$objs scratch/${s}_$tag.o"; else objs="$objs $C/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdg_$tag.so $objs -ldl
echo scratch/libdg_$tag.so
