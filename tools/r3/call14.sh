#!/bin/bash
# round 3, GPU call 14: the row-owned-table step as graph segments around its collectives (world size 1 through RCCL)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c14
timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_fused_gpu.py -q -m gpu -k "parallel or sharded" 2>&1 | grep -E "^E  |Error|passed|failed|FAILED" | head
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-sharded > ${O}_sh_graph.json 2> ${O}_sh_graph.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-sharded --no-graph > ${O}_sh_eager.json 2> ${O}_sh_eager.err
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --force-dp > ${O}_dp.json 2> ${O}_dp.err
for f in sh_graph sh_eager dp; do grep "^{" ${O}_$f.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,2), j['step_us']['median'], j.get('phases'), j['config']['parallelism'], j['config']['hipgraph'])" || tail -5 ${O}_$f.err; done
