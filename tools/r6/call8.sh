#!/bin/bash
# round 6, call 8 (timing experiment, wrong results on purpose): how much of E||D's time is the matrix waves arriving late at the
# row epilogue?  Upper bound: the weight-gradient GEMM loop skipped (nfull = 0), then the hosted election skipped too.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c8; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/base.json 2> $O/base.err
sed -i 's/const int nfull = max(0, r_end - r_begin) \/ span;/const int nfull = 0; r_end_dummy: (void)0;/' deeptables_amd/csrc/deepfm.hip
sed -i 's/for (int base = r_begin + nfull \* span; base < r_end; base += 2) {      \/\/ ragged tail, one K step at a time/for (int base = r_end; base < r_end; base += 2) {/' deeptables_amd/csrc/deepfm.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build1.txt 2>&1; tail -1 $O/build1.txt
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/nogemm.json 2> $O/nogemm.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c8/*.json')):
    try:
        j=[json.loads(l) for l in open(f) if l.startswith('{')][-1]
        print(f.split('/')[-1], round(j['ms_per_step']*1e3,1), 'us', j['step_us'].get('repeat_step_us'), {k:v for k,v in (j.get('kernel_split_us') or {}).items() if k[0] in 'ACEF'})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
