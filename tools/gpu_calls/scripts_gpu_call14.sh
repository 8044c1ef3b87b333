set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest14.log 2>&1
MQ_FUSED_DCN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench14_fused.log 2>&1
MQ_FUSED_DCN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench14_im2col.log 2>&1
tail -3 gpurun_out/pytest14.log
tail -1 gpurun_out/bench14_fused.log | cut -c1-400
tail -1 gpurun_out/bench14_im2col.log | cut -c1-400
