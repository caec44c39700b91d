#!/bin/bash
# steps per hipGraph replay at 200 timed steps (k_prep runs once per replay; a replay boundary idles the GPU a little)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call21.txt
: > $O
run() {
  tag=$1; shift
  python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,3),'M', round(j['ms_per_step']*1e3,2),'us/step median',round(s['median'],2),'p10',round(s['p10'],2),'wall',round(s['wall_us']),'gpu',round(s['gpu_us']),'enq',round(s['host_enqueue_us']))" >> $O
}
run k10 --steps-per-graph 10
run k20 --steps-per-graph 20
run k40 --steps-per-graph 40
run k50 --steps-per-graph 50
run k10b --steps-per-graph 10
run k20_driver --steps 20 --warmup 5 --steps-per-graph 20
run k10_driver --steps 20 --warmup 5 --steps-per-graph 10
cat $O
