# round 2, GPU call 5: VLFuse live-block variants + padding-wave skipping, DyConv fuse with 4 positions in flight, folded text
# operands; full GPU suite; bench with the LVIS-length and the round-1 caption; kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r02_pytest5.log 2>&1; tail -8 gpurun_out/r02_pytest5.log; grep -E "^E  .*(max_err|Error)" gpurun_out/r02_pytest5.log | cut -c1-220 | head -30
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench5_default.log 2>&1; grep -E '^\{' gpurun_out/r02_bench5_default.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 --caption short > gpurun_out/r02_bench5_short.log 2>&1; grep -E '^\{' gpurun_out/r02_bench5_short.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 --batch 16 > gpurun_out/r02_bench5_b16.log 2>&1; grep -E '^\{' gpurun_out/r02_bench5_b16.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 --batch 1 > gpurun_out/r02_bench5_b1.log 2>&1; grep -E '^\{' gpurun_out/r02_bench5_b1.log | cut -c1-200
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-lang-b64 > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_prof5.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r02_prof5 -name "*kernel_trace.csv" -delete; ls gpurun_out/r02_prof5
