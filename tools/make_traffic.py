"""Builds profiles/deepfm_traffic.json (stamped with the source hash of the kernels it was measured on) from the two rocprofv3 PMC passes (tools_pmc.sh):
    python tools/make_traffic.py gpurun_out/r02_pmc_fetch/r02_pmc_fetch_counter_collection.csv \
                                 gpurun_out/r02_pmc_write/r02_pmc_write_counter_collection.csv
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  Correction (MI355X_MICROARCH.md §HBM): on gfx950
FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced STREAMING read -> doubled for the kernels whose
reads are float4 streams; everything else is left as reported (uncalibrated)."""
import collections
import csv
import json
import os
import sys

WIDE = {'k_mlp_fwd3': 'X tile streamed as float4', 'k_tower_x3': 'X tile streamed as float4, weight parts as 16-byte lanes', 'k_dx_sparse_bwd': 'X and dH1 tiles streamed as float4',
        'k_wgrad_rows': 'X / dH1 / H1 / dXn streamed as 8- and 16-byte lanes (its random [m|v] rows are a quarter of its reads)'}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.split('(')[0].replace('void ', '').replace('dt::', '')
    return n.split('<')[0]


def avg_per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and 'dt::' in r['Kernel_Name']:
            acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = avg_per_kernel(sys.argv[1], 'FETCH_SIZE')
    write, _ = avg_per_kernel(sys.argv[2], 'WRITE_SIZE')
    per = {}
    raw = corr = 0.0
    # launches per STEP: kernel A runs once per step; the prep launch only in the first step of a chained execution, the feed
    # gather once per execution, the state's init once per run — each kernel's average is weighted by how often a step meets it
    steps = float(nf.get('k_sparse_fwd', 0) or max(nf.values()))
    for k in fetch:
        f, w = fetch[k], write.get(k, 0.0)
        per_step = min(1.0, nf[k] / steps)
        per[k] = {'FETCH_SIZE': round(f, 1), 'WRITE_SIZE': round(w, 1), 'wide_16B_reads': k in WIDE,
                  'launches_averaged': nf[k], 'launches_per_step': round(per_step, 3)}
        raw += (f + w) * 1024 * per_step
        corr += ((2 * f if k in WIDE else f) + w) * 1024 * per_step
    B, F, D, ND = 8192, 26, 16, 13
    n_dense = (F * D + ND) * 128 + 128 + 128 * 64 + 64 + 64 + 1 + 1 + 2 * (F * D + ND) + (F + ND)
    fwd_bwd = B * (4 * F + 4 * ND + 8 + 12 * F * D) + 12 * n_dense
    opt = B * 6 * 4 * F * D                      # p, m, v of the looked-up rows, read + write
    out = {
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py '
                  '--steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity (tools_pmc.sh), averages per launch weighted by the '
                  'launches per step, KB -> bytes x1024; built by tools/make_traffic.py',
        'step': 'fwd + bwd + Adam (the timed region of bench.py)',
        'per_kernel_KB': per,
        'correction': 'MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) '
                      'coalesced streaming read -> doubled for ' + ', '.join(f'{k} ({v})' for k, v in WIDE.items()) +
                      '; other widths (incl. the 64-byte random row accesses of k_adam_rows_owner / k_sparse_fwd) and '
                      'WRITE_SIZE left as reported (uncalibrated)',
        'bytes_per_step_raw': int(raw),
        'bytes_per_step_corrected': int(corr),
        'algorithmic_bytes_per_step': int(fwd_bwd + opt),
        'algorithmic_fwd_bwd': int(fwd_bwd), 'algorithmic_adam': int(opt),
    }
    out['traffic_over_algorithmic'] = round(out['bytes_per_step_corrected'] / out['algorithmic_bytes_per_step'], 2)
    # counter calibration on known-bytes kernels (tools/make_calibration.py), when given: recorded next to the figures it
    # qualifies — random 64-byte row reads / read-modify-writes and 16-byte-lane streams each have their own factor
    cal = next((a for a in sys.argv[3:] if a.endswith('.json')), None)
    if cal and os.path.exists(cal):
        try:
            out['calibration'] = json.load(open(cal))
        except Exception as e:
            out['calibration'] = {'error': repr(e)}
    # the hash of the kernel sources the passes ran on: taken from the run's own record when given (argv[3] = the
    # bench log holding the "[build] ok ... source hash X" line), else from the tree — bench.py only reports
    # roofline.traffic while this matches the sources it is running
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    out['source_hash'] = ge.source_hash()
    for a in sys.argv[3:]:
        if a.endswith('.log') and os.path.exists(a):
            import re
            m = re.search(r'source hash ([0-9a-f]{16})', open(a).read())
            if m:
                out['source_hash'] = m.group(1)
    path = os.path.join(ROOT, 'profiles', 'deepfm_traffic.json')
    json.dump(out, open(path, 'w'), indent=2)
    print(json.dumps(out, indent=2))


if __name__ == '__main__':
    main()
