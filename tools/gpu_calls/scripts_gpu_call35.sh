mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/vlfuse_check.py 2>&1 | grep -E "FAIL|failures|Error|error" | cut -c1-160
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench35_qb1.log 2>&1; tail -1 gpurun_out/bench35_qb1.log | cut -c1-180
MQ_VLFUSE_QB=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench35_qb2.log 2>&1; tail -1 gpurun_out/bench35_qb2.log | cut -c1-180
