mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r2_t6.log 2>&1
echo "uniform $(grep -h '"metric"' gpurun_out/r2_t6.log | cut -c75-130,190-240)"
