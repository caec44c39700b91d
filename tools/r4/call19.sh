#!/bin/bash
# round 4, call 19: CIN dgrad with two chunks of lookahead: parity + xDeepFM timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c19
O=gpurun_out/r4c19
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_bf16_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "cin or xdeepfm or bf16" > $O/pytest.log 2>&1
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -8
bash tools_prof.sh r4c19_xdeepfm --model xDeepFM --steps 20 --warmup 5 --no-parity > $O/stats_xdeepfm.txt 2>&1
head -8 $O/stats_xdeepfm.txt
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_xdeepfm.json 2> $O/line_xdeepfm.err
python -c "
import json
j=json.loads([l for l in open('$O/line_xdeepfm.json') if l.startswith('{')][-1]); print('xdeepfm ms/step', round(j['ms_per_step'],3), round(j['value']/1e6,3))"
