#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4f; mkdir -p $OUT
for lib in tools/_exp/libsphx_va.so tools/_exp/libsphx_vb.so tools/_exp/libsphx_vc.so; do
  SPHX_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python3 -m pytest tests/test_slab.py -m gpu -q -k "native_ring and (0.01-6.0 or 0.005-12.0)" > $OUT/slab_${lib##*/}.txt 2>&1; echo "lib=[$lib] rc=$?"; tail -3 $OUT/slab_${lib##*/}.txt
done
