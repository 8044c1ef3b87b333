set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tests/gpu_diag.py gpurun_out/diag15.json > gpurun_out/diag15.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench15.log 2>&1
grep -c PASS gpurun_out/diag15.log; grep -E "FAIL|EXCEPTION|Error" gpurun_out/diag15.log | head -40
tail -1 gpurun_out/bench15.log | cut -c1-2600
