mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/dcn_probe.py > gpurun_out/dcn_probe19.txt 2>&1; grep variant gpurun_out/dcn_probe19.txt
timeout 900 python tests/gpu_diag.py gpurun_out/diag19.json > gpurun_out/diag19.log 2>&1
grep -c PASS gpurun_out/diag19.log; grep -E "FAIL|EXCEPTION|Error" gpurun_out/diag19.log | head -20
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench19.log 2>&1
tail -1 gpurun_out/bench19.log | cut -c1-300
