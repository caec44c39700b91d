"""gpurun_out/r06_* (tools/r6/evidence.sh, then tools/r6/refresh.sh on the final source hash) -> profiles/r06_*: the bench lines
(one jsonl, `_run` names the command), the two DeepFM lines as files of their own, the kernel-stats csv per model, the phase
stamps, the MFMA-busy and PMC summaries.  python tools/r6/collect_profiles.py"""
import json
import os
import shutil

G, P = 'gpurun_out', 'profiles'
RUNS = [('driver', 'python bench.py --gpus 1 --steps 20 --warmup 5'), ('deepfm', 'python bench.py'),
        ('zipf', 'python bench.py --dist zipf --no-cpu-baseline'),
        ('deepfm_nochain', 'DT_AMD_CHAIN=0 python bench.py --no-cpu-baseline --no-parity'),
        ('b32768', 'python bench.py --batch 32768 --steps 50 --warmup 10 --no-cpu-baseline'),
        ('b65536', 'python bench.py --batch 65536 --steps 30 --warmup 10 --no-cpu-baseline'),
        ('deepfm_f32tower', 'python bench.py --tower f32 --no-cpu-baseline --no-parity'),
        ('deepfm_bf16tower', 'python bench.py --tower bf16 --no-cpu-baseline'),
        ('dcn', 'python bench.py --model DCN --no-cpu-baseline'),
        ('dcn_dp_w1', 'python bench.py --model DCN --force-dp --no-cpu-baseline --no-parity'),
        ('dp_w1', 'python bench.py --force-dp --no-cpu-baseline --no-parity'),
        ('sharded_w1', 'DT_BENCH_BOTH_LAYOUTS=1 python bench.py --force-sharded --no-cpu-baseline --no-parity'),
        ('xdeepfm', 'python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline'),
        ('autoint', 'python bench.py --model AutoInt --steps 50 --warmup 10 --no-cpu-baseline'),
        ('autoint_f32', 'python bench.py --model AutoInt --attn f32 --steps 50 --warmup 10 --no-cpu-baseline --no-parity'),
        ('autoint_bf16', 'python bench.py --model AutoInt --attn bf16 --steps 50 --warmup 10 --no-cpu-baseline'),
        ('dp_w1_split', 'DT_AMD_DP_GRAPH=0 python bench.py --force-dp --no-cpu-baseline --no-parity')]


def line(name):
    with open(f'{G}/r06_line_{name}.json') as f:
        rows = [l for l in f if l.startswith('{')]
    return json.loads(rows[-1])


def main():
    with open(f'{P}/r06_bench_lines.jsonl', 'w') as out:
        for name, cmd in RUNS:
            j = line(name)
            j['_run'] = cmd
            out.write(json.dumps(j) + '\n')
    for name, dst in (('driver', 'r06_bench_line_driver_command.json'), ('deepfm', 'r06_bench_line.json')):
        with open(f'{P}/{dst}', 'w') as f:
            f.write(json.dumps(line(name)) + '\n')
    for tag in ('deepfm', 'deepfm_zipf', 'deepfm_b32768', 'deepfm_b65536', 'dcn', 'autoint', 'xdeepfm'):
        shutil.copy(f'{G}/r06_{tag}/r06_{tag}_kernel_stats.csv', f'{P}/r06_{tag}_kernel_stats.csv')
    shutil.copy(f'{G}/r06_deepfm_phase_stamps.txt', f'{P}/r06_deepfm_phase_stamps.txt')
    shutil.copy(f'{G}/r06_pmc_mfma.txt', f'{P}/r06_mfma_busy.txt')
    shutil.copy(f'{G}/r06_pmc_fetch.txt', f'{P}/r06_deepfm_pmc_fetch_size.txt')
    shutil.copy(f'{G}/r06_pmc_write.txt', f'{P}/r06_deepfm_pmc_write_size.txt')
    shutil.copy(f'{G}/deepfm_traffic.json', f'{P}/deepfm_traffic.json')
    for name, _ in RUNS:
        j = line(name)
        print(f"{name:18s} {j['value'] / 1e6:8.3f} M rows/s  {j['ms_per_step'] * 1e3:8.1f} us  median {j['step_us']['median']:7.1f}  "
              f"parity {(j.get('parity') or {}).get('ok')}")


if __name__ == '__main__':
    main()
