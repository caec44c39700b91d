#!/bin/bash
# round 6, call 16: F without tickets (real implementation): parity tests + lines; then ablations inside kernel C's dXn phase
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c16; mkdir -p $O
TW=deeptables_amd/csrc/tower_x3.h
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_headline_gpu.py tests/test_compiled_gpu.py tests/test_optim_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver.json 2> $O/driver.err
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/base200.json 2> $O/base200.err
cp $TW /tmp/tower.orig
run() {
  python -c "import __graft_entry__ as g; g.build()" > $O/build_$1.txt 2>&1 || { echo "$1 build failed"; tail -3 $O/build_$1.txt; return; }
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
}
# C3: dXn phase without its two BN-backward atomics per column
sed -i 's/            if (kg == 0 \&\& col < dm.C) { radd(prec + pl.sdx + col, s1); radd(prec + pl.sdxx + col, s2); }/            if (kg == 0 \&\& col < dm.C \&\& s1 == 12345.f) { radd(prec + pl.sdx + col, s1); radd(prec + pl.sdxx + col, s2); }/' $TW; run c_no_sdx_atomics; cp /tmp/tower.orig $TW
# C4: dXn phase without the six-MFMA groups (loads and epilogue stay)
sed -i 's/                X3_MFMA(gH\[0\], ah\[0\]\[g\], bW\[buf\]\[g\]\[0\]); X3_MFMA(gH\[1\], ah\[1\]\[g\], bW\[buf\]\[g\]\[0\]);/                gH[0][0] += (float)bW[buf][g][0][0]; gH[1][0] += (float)bW[buf][g][1][0];/; s/                X3_LO(gM\[0\], ah\[0\]\[g\], bW\[buf\]\[g\]\[1\]); X3_LO(gM\[1\], ah\[1\]\[g\], bW\[buf\]\[g\]\[1\]);/                gM[0][0] += (float)ah[0][g][0] + (float)al[0][g][0];/; s/                X3_LO(gM\[0\], al\[0\]\[g\], bW\[buf\]\[g\]\[0\]); X3_LO(gM\[1\], al\[1\]\[g\], bW\[buf\]\[g\]\[0\]);/                gM[1][0] += (float)ah[1][g][0] + (float)al[1][g][0];/' $TW; run c_no_dxn_mfma; cp /tmp/tower.orig $TW
python - <<'PY'
import json
for f in ['driver','base200','c_no_sdx_atomics','c_no_dxn_mfma']:
    try:
        j=[json.loads(l) for l in open(f'gpurun_out/r6c16/{f}.json') if l.startswith('{')][-1]
        print(f'{f:18s}', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,2), j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f'gpurun_out/r6c16/{f}.err').read()[-300:])
PY
