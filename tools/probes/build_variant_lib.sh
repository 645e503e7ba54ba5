#!/bin/bash
# Build a MEASUREMENT variant of libsphx.so from the working tree with extra -D flags into tools/_exp/ (git-ignored; travels to
# the GPU box), e.g.   tools/probes/build_variant_lib.sh pretend320 -DSPHX_EXP_PRETEND_COMPLETE_TILE
# Use it through SPHX_LIB=tools/_exp/libsphx_<tag>.so or the "@lib" form of tools/probes/probe_ab_switches.py.
set -e
TAG=${1:?tag}; shift
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
C="$ROOT/sph-poiseuille-flow_amd/csrc"; W=$(mktemp -d)
for s in sphx_common sphx_pairlist sphx_resident; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -c "$C/$s.hip" -o "$W/$s.o" &
done; wait
mkdir -p "$ROOT/tools/_exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_exp/libsphx_$TAG.so" "$W"/*.o
rm -rf "$W"; echo "$ROOT/tools/_exp/libsphx_$TAG.so"
