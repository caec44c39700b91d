#!/bin/bash
# the bench's own host overhead inside the 20-step region: timing events created before it; the host polling the last
# event instead of sleeping on the interrupt.  Same box, same binary, three repeats each (host noise is +-100 us)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call18.txt
: > $O
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,2),'M', round(j['ms_per_step']*1e3,1),'us/step  wall',round(s['wall_us']),'gpu',round(s['gpu_us']),'enq',round(s['host_enqueue_us']),'e0',round(s['first_record_us']),'done',round(s['host_done_us']),'units',s.get('unit_us'))" >> $O
}
for r in 1 2 3; do
run old DT_BENCH_NO_EVENT_POOL=1 --   # (the switches of this A/B were removed from bench.py after the call: the pool stayed, the poll went)
run pool X=1 --
run pool_poll DT_BENCH_POLL=1 --
run pool_poll_nointr DT_BENCH_POLL=1 HSA_ENABLE_INTERRUPT=0 --
done
cat $O
