#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c10; mkdir -p $O
timeout 900 python tools/r6/dbg_eager_repeat.py > $O/dbg.txt 2>&1
tail -6 $O/dbg.txt
