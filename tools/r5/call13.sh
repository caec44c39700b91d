#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/op_profile_step.py xDeepFM > gpurun_out/c13_ops_xdeepfm.txt 2>&1
timeout 300 python tools/op_profile_step.py AutoInt > gpurun_out/c13_ops_autoint.txt 2>&1
grep "us/step" gpurun_out/c13_ops_xdeepfm.txt | head -45
echo ======
grep "us/step" gpurun_out/c13_ops_autoint.txt | head -40
