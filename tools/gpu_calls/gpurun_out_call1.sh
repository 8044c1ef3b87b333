set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0)); import os; print('cores', os.cpu_count())" > gpurun_out/env.log 2>&1
rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -8 >> gpurun_out/env.log 2>&1
timeout 900 python tests/gpu_diag.py gpurun_out/diag1.json > gpurun_out/diag1.log 2>&1
echo "diag exit $?" >> gpurun_out/diag1.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench1.log 2>&1
echo "bench exit $?" >> gpurun_out/bench1.log
tail -5 gpurun_out/diag1.log; tail -3 gpurun_out/bench1.log
