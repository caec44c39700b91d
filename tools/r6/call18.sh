#!/bin/bash
# round 6, call 18: which Python lines launch the small torch kernels (fills / copies / adds) of the AutoInt and xDeepFM steps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c18; mkdir -p $O
timeout 600 python tools/r6/glue_trace.py AutoInt 3 > $O/autoint.txt 2>&1
timeout 600 python tools/r6/glue_trace.py xDeepFM 3 > $O/xdeepfm.txt 2>&1
head -60 $O/autoint.txt; echo; head -40 $O/xdeepfm.txt
