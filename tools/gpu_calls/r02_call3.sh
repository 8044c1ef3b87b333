# round 2, GPU call 3: Swin MLP v2, grouped DyConv coefficients, MSDeformAttn, MQ-GLIP-L family (window 12), microbenchmarks,
# MQ-GLIP-L bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r02_pytest3.log 2>&1; tail -8 gpurun_out/r02_pytest3.log; grep -E "^E  .*(max_err|Error)" gpurun_out/r02_pytest3.log | cut -c1-220 | head -30
timeout 300 python tools/microbench.py gpurun_out/r02_microbench3.json 2>&1 | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-lang-b64 > gpurun_out/r02_bench3_default.log 2>&1; grep -E '^\{' gpurun_out/r02_bench3_default.log | cut -c1-300
timeout 600 python bench.py --workload mq-glip-l --steps 5 > gpurun_out/r02_bench3_glipl.log 2>&1; grep -E '^\{' gpurun_out/r02_bench3_glipl.log | cut -c1-400; tail -3 gpurun_out/r02_bench3_glipl.log | cut -c1-300
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-lang-b64 > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_prof3.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r02_prof3 -name "*kernel_trace.csv" -delete; ls gpurun_out/r02_prof3
