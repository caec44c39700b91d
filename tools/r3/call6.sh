#!/bin/bash
# round 3, GPU call 6: how much of the step is the fixed cost of a hipGraph replay?  K train steps per captured graph
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c6
for k in 1 2 5 10; do
  timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --steps-per-graph $k > ${O}_line_k$k.json 2> ${O}_line_k$k.err
  echo K=$k; python - <<PY
import json
j=json.loads(open('${O}_line_k$k.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['step_us'], j['config'].get('steps_per_graph_replay'))
PY
done
timeout 400 bash tools_prof.sh r3c6_prof_k5 --steps 100 --warmup 10 --no-parity --steps-per-graph 5 > ${O}_stats_k5.txt 2>&1
head -8 ${O}_stats_k5.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3c6_prof_k5/r3c6_prof_k5_kernel_trace.csv')))
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:30]) for r in rows)
n=len(ks); prev=None
for s,e,name in ks[n-45:n-5]:
    gap=(s-prev)/1e3 if prev else 0
    if gap>0.5: print(f'{name:32s} dur {(e-s)/1e3:7.2f}  gap_before {gap:6.2f}')
    prev=e
PY
