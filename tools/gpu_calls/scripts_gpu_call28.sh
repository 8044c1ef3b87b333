mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest28.log 2>&1; tail -2 gpurun_out/pytest28.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench28.log 2>&1
tail -1 gpurun_out/bench28.log | cut -c1-300
