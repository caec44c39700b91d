#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6suite; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.txt 2>&1
tail -15 $O/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
