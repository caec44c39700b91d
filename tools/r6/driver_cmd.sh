#!/bin/bash
# the driver's command on a fresh box with a multi-second idle in front (VERDICT r5 #1e); one line per call -> profiles/r06_driver_command_lines.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6drv; mkdir -p $O
N=${1:-x}
sleep 8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_$N.json 2> $O/line_$N.err
python - <<PY
import json
j=json.loads(open('$O/line_$N.json').read().strip().splitlines()[-1])
su=j['step_us']
print('value', round(j['value']/1e6,2), 'M rows/s  ms_per_step', round(j['ms_per_step']*1e3,1), 'us; repeats', su.get('repeat_step_us'), 'steady', round(j.get('steady_rows_per_s',0)/1e6,2), 'split', j.get('kernel_split_us'), 'clocks', (su.get('clocks') or {}).get('contract_region'), 'var', {k:(round(v['rows_per_s']/1e6,1)) for k,v in (j.get('variants') or {}).items()}, 'parity', j['parity']['ok'], 'cpu', j.get('cpu_baseline'))
PY
