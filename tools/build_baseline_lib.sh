#!/bin/bash
# Build libsphx.so of an earlier commit into tools/_exp/ (git-ignored) for A/B runs on one box: SPHX_LIB=tools/_exp/libsphx_<tag>.so
#   tools/build_baseline_lib.sh e4817a8 r2     (round 2's final library, what tools/r3_ab.sh compares against)
set -e
REV=${1:?commit}; TAG=${2:?tag}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WT=$(mktemp -d)
git -C "$ROOT" worktree add -q "$WT" "$REV"
(cd "$WT" && python3 -c "
import importlib, sys
sys.path.insert(0, '.')
print(importlib.import_module('sph-poiseuille-flow_amd.build').build())")
mkdir -p "$ROOT/tools/_exp"
cp "$WT/sph-poiseuille-flow_amd/csrc/libsphx.so" "$ROOT/tools/_exp/libsphx_$TAG.so"
git -C "$ROOT" worktree remove --force "$WT"
echo "$ROOT/tools/_exp/libsphx_$TAG.so"
