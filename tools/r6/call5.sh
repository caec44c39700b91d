#!/bin/bash
# round 6, call 5: fresh-box driver command (#3); bf16 attention (two-part forward) tests + lines; then the BN-shard A/B (8 vs 4)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c5; mkdir -p $O gpurun_out/r6drv
sleep 8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6drv/line_3.json 2> gpurun_out/r6drv/line_3.err
timeout 900 python -m pytest tests/test_autoint_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --model AutoInt --attn bf16 > $O/autoint_bf16.json 2> $O/autoint_bf16.err
# A/B: BatchNormalization batch-sum shards 8 -> 4 (kernel C's prologue loads halve; kernel A's atomics meet 128 deep)
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/shards8.json 2> $O/shards8.err
sed -i 's/^constexpr int kBnShards = 8;/constexpr int kBnShards = 4;/' deeptables_amd/csrc/deepfm.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build4.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/shards4.json 2> $O/shards4.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/shards4b.json 2> $O/shards4b.err
python - <<'PY'
import json,glob
for f in ['gpurun_out/r6drv/line_3.json']+sorted(glob.glob('gpurun_out/r6c5/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        su=j['step_us']
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), 'M; rep', su.get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), 'split', {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
        p=j.get('parity')
        if p and not p.get('ok'):
            u=p.get('uniform') or p.get('zipf') or {}
            print('   ', {k:u[k] for k in u if 'err' in k})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
