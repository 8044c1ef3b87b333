# round 2, GPU call 1: all GPU tests (new: benchmark-configuration parity, reference-kernel pins, golden post-processing,
# caches), error ladder with the fp16-operand floor, bench (new caption, honest rooflines, B=64 language path, config-1 CPU
# baseline), fp16-stream A/B, LVIS-protocol workload, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r02_pytest1.log 2>&1; tail -25 gpurun_out/r02_pytest1.log
timeout 600 python tests/gpu_diag.py --ladder gpurun_out/r02_error_ladder.txt > gpurun_out/r02_ladder1.log 2>&1; tail -5 gpurun_out/r02_ladder1.log
( time timeout 900 python bench.py ) > gpurun_out/r02_bench1_default.log 2>&1; grep -E '^\{|real' gpurun_out/r02_bench1_default.log | cut -c1-400
MQ_RESIDUAL_FP32=0 timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 > gpurun_out/r02_bench1_fp16streams.log 2>&1; grep -E '^\{' gpurun_out/r02_bench1_fp16streams.log | cut -c1-200
timeout 400 python bench.py --workload lvis --batch 1 --steps 4 --warmup 2 > gpurun_out/r02_bench1_lvis_b1.log 2>&1; grep -E '^\{' gpurun_out/r02_bench1_lvis_b1.log | cut -c1-300
timeout 400 python bench.py --workload lvis --batch 8 --steps 2 --warmup 2 > gpurun_out/r02_bench1_lvis_b8.log 2>&1; grep -E '^\{' gpurun_out/r02_bench1_lvis_b8.log | cut -c1-300
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-lang-b64 > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_prof1.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r02_prof1 -name "*kernel_trace.csv" -delete; ls gpurun_out/r02_prof1
