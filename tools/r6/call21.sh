#!/bin/bash
# round 6, call 21: kernel table of the AutoInt step with the head on the pending normalisation
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools_prof.sh r6c21_autoint --steps 100 --warmup 20 --model AutoInt --no-parity | cut -c1-200
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6c21_autoint/r6c21_autoint_kernel_stats.csv')))
for r in rows[14:32]:
    print(f"{float(r['AverageNs'])/1e3:9.2f}us x{r['Calls']:>5} {float(r['Percentage']):5.1f}%  {r['Name'][:90]}")
PY
