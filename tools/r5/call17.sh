#!/bin/bash
# where do the ~200 us the driver's 20-step command pays over 20 x the steady step go?  same box, same binary:
# steps per replay, host wait mode, collector before the warm-up instead of after it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call17.txt
: > $O
run() { # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,2),'M', round(j['ms_per_step']*1e3,1),'us/step  wall',round(s['wall_us']),'gpu',round(s['gpu_us']),'enq',round(s['host_enqueue_us']),'units',s.get('unit_us'))" >> $O
}
run base1 X=1 --
run gc_early DT_BENCH_GC_EARLY=1 --
run nointr HSA_ENABLE_INTERRUPT=0 --
run nointr_gc HSA_ENABLE_INTERRUPT=0 DT_BENCH_GC_EARLY=1 --
run spg5 X=1 -- --steps-per-graph 5
run spg4 X=1 -- --steps-per-graph 4
run spg2 X=1 -- --steps-per-graph 2
run spg20 X=1 -- --steps-per-graph 20
run warm10 X=1 -- --warmup 10
run base2 X=1 --
cat $O
