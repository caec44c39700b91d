#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/r5/dbg_elect.py > gpurun_out/c3_dbg.txt 2>&1
tail -60 gpurun_out/c3_dbg.txt
bash tools_prof.sh c3_b8192 --steps 100 --warmup 10 --no-parity | head -8
