#!/bin/bash
# layer-by-layer models with all dense weights / gradients in one flat buffer (DT_AMD_FLAT_PARAMS=1: off by default since the
# DCN layer-by-layer step measured slower with it) — xDeepFM and AutoInt, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_call20.txt
: > $O
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['step_us']
print('$tag', round(j['value']/1e6,3),'M', round(j['ms_per_step']*1e3,1),'us/step median',round(s['median'],1),'parity',(j.get('parity') or {}).get('ok'))" >> $O
}
run xdeepfm X=1 -- --model xDeepFM --steps 40 --warmup 10
run xdeepfm_flat DT_AMD_FLAT_PARAMS=1 -- --model xDeepFM --steps 40 --warmup 10
run autoint X=1 -- --model AutoInt --steps 50 --warmup 10
run autoint_flat DT_AMD_FLAT_PARAMS=1 -- --model AutoInt --steps 50 --warmup 10
cat $O
