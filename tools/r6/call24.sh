#!/bin/bash
# round 6, call 24: A/B on one box — the tile kernel's record shards (kRecShards 4 / 8 / 2): its 864 double atomics per tile
# were priced at 2.8 us of the launch (call 16); more shards = shorter same-address queues, more loads in E||D / F
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c24; mkdir -p $O
run() {
  tag=$1
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/drv_$tag.json 2> $O/drv_$tag.err
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-extras > $O/d200_$tag.json 2> $O/d200_$tag.err
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
for f in ['drv','d200']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c24/{f}_{tag}.json') if l.startswith('{')][-1]
        print(tag, f, round(j['ms_per_step']*1e3,1), 'us', j['step_us'].get('repeat_step_us'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(tag, f, 'ERR', e)
PY
}
run s4a
for n in 8 2; do
  sed -i "s/^constexpr int kRecShards = [0-9]*;/constexpr int kRecShards = $n;/" deeptables_amd/csrc/deepfm.hip
  python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
  run s$n
done
sed -i "s/^constexpr int kRecShards = [0-9]*;/constexpr int kRecShards = 4;/" deeptables_amd/csrc/deepfm.hip
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run s4b
