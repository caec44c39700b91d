#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3c8
for cfg in "1 512" "5 512" "1 1024" "5 1024" "5 2048"; do
  set -- $cfg
  DT_ADAM_SEG_BLOCKS=$2 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extras --steps 200 --dist zipf --steps-per-graph $1 > ${O}_z_$1_$2.json 2> ${O}_z_$1_$2.err
  echo "K=$1 seg=$2"; python - <<PY
import json
j=json.loads(open('${O}_z_$1_$2.json').read().strip().splitlines()[-1])
print(j['value'], j['step_us'])
PY
done
timeout 400 bash tools_prof.sh r3c8_prof_zipf --steps 100 --warmup 10 --no-parity --dist zipf > ${O}_stats_zipf.txt 2>&1
head -8 ${O}_stats_zipf.txt
