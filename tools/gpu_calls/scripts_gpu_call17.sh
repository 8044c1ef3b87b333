mkdir -p gpurun_out
for v in 0 1 2 3 4; do MQ_DCN_VARIANT=$v timeout 120 python tools/dcn_probe.py; done > gpurun_out/dcn_probe.txt 2>&1
cat gpurun_out/dcn_probe.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench17.log 2>&1
tail -1 gpurun_out/bench17.log | cut -c1-300
