#!/bin/bash
# round 6, call 15 (timing ablation): F without the arrival tickets (the state advanced by one thread, unordered)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c15; mkdir -p $O
SRC=deeptables_amd/csrc/deepfm.hip
cp $SRC /tmp/deepfm.orig
run() {
  python -c "import __graft_entry__ as g; g.build()" > $O/build_$1.txt 2>&1 || { echo "$1 build failed"; tail -3 $O/build_$1.txt; return; }
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
}
sed -i '2271s/adam_finish(st, threadIdx.x == 0 ? 0u : kNoTicket, lr, da.b1, da.b2);/if (blockIdx.x == 0 \&\& threadIdx.x == 0 \&\& st) { const int t = st->t + 1; st->t = t; st->lr_t = adam_lr_t(lr, da.b1, da.b2, t); }/' $SRC
sed -n 2271p $SRC
run f_no_tickets
sed -i '2260s/if (b < col_blocks) {/if (false) {/; 2262s/} else if (b < col_blocks + small_blocks) {/} else if (false) {/; 2266s/} else if (fs.seg.nseg) {/} else if (false) {/' $SRC; run f_empty_no_tickets; cp /tmp/deepfm.orig $SRC
python - <<'PY'
import json
for f in ['f_no_tickets','f_empty_no_tickets']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c15/{f}.json') if l.startswith('{')][-1]
        print(f'{f:18s}', round(j['ms_per_step']*1e3,1), 'us', j['step_us'].get('repeat_step_us'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c15/{f}.err').read()[-300:])
PY
