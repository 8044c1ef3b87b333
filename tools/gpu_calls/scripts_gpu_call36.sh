mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/dcn_check.py 2>&1 | grep -E "FAIL|failures|Error|error" | cut -c1-160
timeout 100 python tools/dcn_probe.py 2>&1 | grep variant
MQ_DCN_WAVES=8 timeout 100 python tools/dcn_probe.py 2>&1 | grep variant
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench36_w16.log 2>&1; tail -1 gpurun_out/bench36_w16.log | cut -c1-180
