#!/bin/bash
# which lines issue the small torch ops of the layer-by-layer steps (xDeepFM, AutoInt)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/op_shapes_step.py xDeepFM > gpurun_out/c19_shapes_xdeepfm.txt 2>&1
timeout 200 python tools/op_shapes_step.py AutoInt > gpurun_out/c19_shapes_autoint.txt 2>&1
grep -c . gpurun_out/c19_shapes_xdeepfm.txt gpurun_out/c19_shapes_autoint.txt
grep "^aten" gpurun_out/c19_shapes_xdeepfm.txt | head -70
