mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest37.log 2>&1; tail -3 gpurun_out/pytest37.log
( time timeout 600 python bench.py ) > gpurun_out/bench37_default.log 2>&1; grep -E '^\{|real' gpurun_out/bench37_default.log | cut -c1-220
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof37 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof37.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof37/bench_kernel_trace.csv; ls gpurun_out/prof37
