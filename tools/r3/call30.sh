#!/bin/bash
# round 3, GPU call 30: the three N > 1 step-structure lines again with the final bench.py (roofline.traffic null there)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r03
python bench.py --force-dp --no-cpu-baseline --no-parity > ${O}_line_dp_w1.json 2> ${O}_line_dp_w1.err
python bench.py --force-dp --dist zipf --no-cpu-baseline --no-parity > ${O}_line_dp_w1_zipf.json 2> ${O}_line_dp_w1_zipf.err
python bench.py --force-sharded --no-cpu-baseline --no-parity > ${O}_line_sharded_w1.json 2> ${O}_line_sharded_w1.err
for f in dp_w1 dp_w1_zipf sharded_w1; do grep "^{" ${O}_line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e6,3), round(j['ms_per_step']*1e3,1), j['roofline']['traffic'], j.get('phases'))"; done
