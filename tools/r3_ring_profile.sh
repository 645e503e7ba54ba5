#!/bin/bash
# kernel trace of an in-process slab ring running on ONE stream (serial: per-kernel durations = one slab alone on the chip)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=${1:-C5}; G=${2:-8}; STEPS=${3:-40}; TAG=${4:-a}
O=gpurun_out/r3_ring_$TAG
mkdir -p $O
python tools/probes/probe_slab_ring.py $WL $G $STEPS one-stream | tee $O/ring.txt
python tools/probes/probe_slab_ring.py $WL $G $STEPS | tee -a $O/ring.txt
rocprofv3 --kernel-trace -d $O/prof -o t -- python3 tools/probes/probe_slab_ring.py $WL $G $STEPS one-stream > $O/prof.log 2>&1
python tools/rocprof_kernels.py $(ls $O/prof/*.db | head -1) 45 | tee $O/kernels.txt
