#!/bin/bash
# round 4, call 13: where the host clock of the driver's 20-step region goes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c13
O=gpurun_out/r4c13
for i in 1; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-parity --no-cpu-baseline --no-extras > $O/line_$i.json 2> $O/line_$i.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_nospin.json 2> $O/line_nospin.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_full.json 2> $O/line_full.err
for f in 1 nospin full; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    s=j['step_us']
    print('$f', round(j['value']/1e6,2),'M rows/s', 'wall', round(s['wall_us']), 'gpu', round(s['gpu_us']), 'enq', round(s['host_enqueue_us']), 'median', round(s['median'],1), 'first', j.get('first_replay_us'))
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-800:])
PY
done
