#!/bin/bash
# the first ~180 steps after an idle period run 4 % slower (call 22) even behind 400 warm-up steps: is it the idle time of the
# collector pass between warm-up and timed region?  collector first, 60 ms of streaming work, then barrier + synchronize + go
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call23.txt
: > $O
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,3),'M', round(j['ms_per_step']*1e3,2),'us/step wall',round(s['wall_us']),'gpu',round(s['gpu_us']),'units',[round(u/j['config']['steps_per_graph_replay'],1) for u in s.get('unit_us',[])])" >> $O
}
run drv X=1 -- --steps 20 --warmup 5
run drv_preheat DT_BENCH_PREHEAT=1 -- --steps 20 --warmup 5    # (switch removed from bench.py after the call: 107.0 -> 106.1-106.4 us, not worth it)
run s200_preheat DT_BENCH_PREHEAT=1 -- --no-parity
run drv_preheat2 DT_BENCH_PREHEAT=1 -- --steps 20 --warmup 5
cat $O
