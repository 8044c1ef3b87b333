mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tests/gpu_diag.py gpurun_out/diag31.json > gpurun_out/diag31.log 2>&1
grep -c PASS gpurun_out/diag31.log; grep -E "FAIL|EXCEPTION|Error" gpurun_out/diag31.log | head -20
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest31.log 2>&1; tail -2 gpurun_out/pytest31.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench31.log 2>&1
tail -1 gpurun_out/bench31.log | cut -c1-300
