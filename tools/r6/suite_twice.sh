#!/bin/bash
# round 6: flakiness check — the full GPU suite twice on one fresh box (the driver runs it with -x at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6suite; mkdir -p $O
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/run_$1_$i.txt 2>&1
  tail -3 $O/run_$1_$i.txt
done
