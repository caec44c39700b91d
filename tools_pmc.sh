#!/bin/bash
# usage: tools_pmc.sh <tag> "<counters>" [bench args]  -> per-kernel averaged counters
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/$tag -o $tag -- python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
python - <<PY
import csv, collections
f='gpurun_out/$tag/${tag}_counter_collection.csv'
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if not k.startswith('dt::') and 'dt::' not in k: continue
    print(k)
    for c,vals in v.items(): print(f'    {c:28s} avg={sum(vals)/len(vals):14.1f} n={len(vals)}')
PY
