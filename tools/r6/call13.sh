#!/bin/bash
# round 6, call 13 (timing ablations, wrong results on purpose; `cp` restores the source between them): what each piece of the
# step's launches costs the STEP — the method that found the weight-gradient lever (call 8)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c13; mkdir -p $O
SRC=deeptables_amd/csrc/deepfm.hip
cp $SRC /tmp/deepfm.orig
run() {  # name
  python -c "import __graft_entry__ as g; g.build()" > $O/build_$1.txt 2>&1 || { echo "$1 build failed"; tail -3 $O/build_$1.txt; return; }
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
}
run base
# E1: no hosted election in E||D
sed -i '2453s/if (nx.idx) {/if (false \&\& nx.idx) {/' $SRC; run no_election; cp /tmp/deepfm.orig $SRC
# E4a: no segment walk in F
sed -i '2266s/} else if (fs.seg.nseg) {/} else if (false) {/' $SRC; run no_segwalk; cp /tmp/deepfm.orig $SRC
# E4b: no column blocks in F (slices -> dW1 / dW2, dense Adam, layouts)
sed -i '2260s/if (b < col_blocks) {/if (false) {/' $SRC; run no_colblocks; cp /tmp/deepfm.orig $SRC
# E5a: no BN batch-sum atomics in A
sed -i '321s/unsafeAtomicAdd(acc + col, sum);/;/; 322s/unsafeAtomicAdd(acc + dm.CP + col, sq);/;/' $SRC; run no_bnatomics; cp /tmp/deepfm.orig $SRC
# E5b: A does not pack the next step's rows
sed -i '260s/if (nx.idx) {/if (false) {/' $SRC; run no_nextrows; cp /tmp/deepfm.orig $SRC
python - <<'PY'
import json,glob
for f in ['base','no_election','no_segwalk','no_colblocks','no_bnatomics','no_nextrows']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c13/{f}.json') if l.startswith('{')][-1]
        print(f'{f:14s}', round(j['ms_per_step']*1e3,1), 'us', j['step_us'].get('repeat_step_us'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c13/{f}.err').read()[-300:])
PY
