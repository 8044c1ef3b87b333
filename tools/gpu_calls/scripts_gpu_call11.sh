set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/gpu_diag.py gpurun_out/diag11.json > gpurun_out/diag11.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest11.log 2>&1
MQ_FUSED_DCN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench11_fused.log 2>&1
MQ_FUSED_DCN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench11_im2col.log 2>&1
grep -c PASS gpurun_out/diag11.log; grep -E "FAIL|EXCEPTION" gpurun_out/diag11.log; tail -3 gpurun_out/pytest11.log
tail -1 gpurun_out/bench11_fused.log | cut -c1-1500; tail -1 gpurun_out/bench11_im2col.log | cut -c1-1500
