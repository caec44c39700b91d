#!/bin/bash
# the replays of a 200-step run get faster from first to last (2118 -> 2018 us per twenty steps): where does it settle?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call22.txt
: > $O
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-parity --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,3),'M', round(j['ms_per_step']*1e3,2),'us/step median',round(s['median'],2),'p10',round(s['p10'],2),'units',[round(u/j['config']['steps_per_graph_replay'],1) for u in s.get('unit_us',[])])" >> $O
}
run s1000 X=1 -- --steps 1000 --warmup 20
run s200_spin DT_BENCH_SPIN=1 -- --steps 200 --warmup 20
run s200_warm400 X=1 -- --steps 200 --warmup 400
cat $O
