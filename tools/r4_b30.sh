#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4x; mkdir -p $OUT
(timeout -k 10 60 tools/_exp/bench_graph_fork 300 8 40 && timeout -k 10 60 tools/_exp/bench_graph_fork 80 8 40 && timeout -k 10 60 tools/_exp/bench_graph_fork 300 3 40) 2>&1 | tee $OUT/graph_fork.txt
