#!/bin/bash
# round 6, call 1: what the box offers for clocks; the driver's command on the new bench.py (warm-up through the graph,
# repeats, clocks, kernel split) against round 5's bench.py on the same box, alternating; the chained / compiled tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c1; mkdir -p $O
{
  echo "== sysfs"; ls /sys/class/drm/ 2>&1; for c in /sys/class/drm/card*/device; do echo $c; ls $c | tr '\n' ' '; echo; cat $c/pp_dpm_sclk $c/pp_dpm_mclk $c/pp_dpm_fclk $c/power_dpm_force_performance_level 2>&1; ls $c/hwmon/* 2>&1 | tr '\n' ' '; done
  echo "== rocm-smi"; rocm-smi --showclocks --showpower --showperflevel --showmemuse 2>&1 | head -60
  echo "== amdsmi"; python -c "import amdsmi; print(amdsmi.__file__)" 2>&1 | tail -1
  rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20
  nproc; free -g | head -2
} > $O/probe.txt 2>&1
sleep 5
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/new1.json 2> $O/new1.err
sleep 5
python tools/r6/bench_r05.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/old1.json 2> $O/old1.err
sleep 5
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/new2.json 2> $O/new2.err
sleep 5
python tools/r6/bench_r05.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/old2.json 2> $O/old2.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legacy-warmup > $O/new_legacy.json 2> $O/new_legacy.err
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_compiled_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c1/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        su=j['step_us']
        print(f, round(j['ms_per_step']*1e3,1), 'us  rep', su.get('repeat_step_us'), 'pause', su.get('pause_before_region_us'), 'clk', (su.get('clocks') or {}).get('before_region'), 'split', j.get('kernel_split_us'), 'var', j.get('variants'))
    except Exception as e:
        print(f, 'ERR', e)
PY
