#!/bin/bash
# round 3, GPU call 28: the whole GPU suite + smoke() at the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c28
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 > ${O}_tests.txt
grep -E "passed|failed|FAILED" ${O}_tests.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench default-ish', round(j['value']/1e6,3), j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['parity']['ok'], j['config']['steps_per_graph_replay'])"
