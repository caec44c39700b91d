#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   -> gpurun_out/<tag>/ (rocprofv3 kernel stats csv) + gpurun_out/<tag>.log
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o $tag -- python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$tag/${tag}_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.2f}us x{r['Calls']:>5} {float(r['Percentage']):5.1f}%  {r['Name'][:90]}")
PY
grep '"metric"' gpurun_out/$tag.log | cut -c1-330
