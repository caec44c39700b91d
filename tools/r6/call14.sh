#!/bin/bash
# round 6, call 14 (timing ablations, wrong results on purpose): the floors of F and C
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c14; mkdir -p $O
SRC=deeptables_amd/csrc/deepfm.hip; TW=deeptables_amd/csrc/tower_x3.h
cp $SRC /tmp/deepfm.orig; cp $TW /tmp/tower.orig
run() {
  python -c "import __graft_entry__ as g; g.build()" > $O/build_$1.txt 2>&1 || { echo "$1 build failed"; tail -3 $O/build_$1.txt; return; }
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
}
# F1: F with no role at all (tickets only)
sed -i '2260s/if (b < col_blocks) {/if (false) {/; 2262s/} else if (b < col_blocks + small_blocks) {/} else if (false) {/; 2266s/} else if (fs.seg.nseg) {/} else if (false) {/' $SRC; run f_tickets_only; cp /tmp/deepfm.orig $SRC
# F2: ... and 512 instead of 1024 segment blocks
sed -i 's/const int seg_blocks = 1024;/const int seg_blocks = 512;/' $SRC; run f_seg512; cp /tmp/deepfm.orig $SRC
# C1: the tile kernel without its dXn GEMM phase
sed -i 's/        if (wave < NT) tile(0, wave);/        if (false) tile(0, wave);/; s/        if (wave + 8 < NT) tile(1, wave + 8);/        if (false) tile(1, wave + 8);/; s/        if (wave + 16 < NT) tile(0, wave + 16);/        if (false) tile(0, wave + 16);/; s/        if (wave + 24 < NT) tile(1, wave + 24);/        if (false) tile(1, wave + 24);/' $TW; run c_no_dxn; cp /tmp/tower.orig $TW
# C2: the tile kernel without the batch-sum loads of its prologue
sed -i 's/        bsx\[w\] = p.bnacc\[(int64_t)w \* 2 \* dm.CP + bcol\];/        bsx[w] = 1.0;/; s/        bsq\[w\] = p.bnacc\[(int64_t)w \* 2 \* dm.CP + dm.CP + bcol\];/        bsq[w] = 2.0;/' $TW; run c_no_bnloads; cp /tmp/tower.orig $TW
python - <<'PY'
import json
for f in ['f_tickets_only','f_seg512','c_no_dxn','c_no_bnloads']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c14/{f}.json') if l.startswith('{')][-1]
        print(f'{f:16s}', round(j['ms_per_step']*1e3,1), 'us', j['step_us'].get('repeat_step_us'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c14/{f}.err').read()[-300:])
PY
