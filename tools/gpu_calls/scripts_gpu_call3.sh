set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/gpu_diag.py gpurun_out/diag3.json > gpurun_out/diag3.log 2>&1
echo "diag exit $?" >> gpurun_out/diag3.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench3.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof3.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof3 | head
( time timeout 250 python bench.py --cpu-baseline-worker ) > gpurun_out/cpu3.log 2>&1
grep -c PASS gpurun_out/diag3.log; grep FAIL gpurun_out/diag3.log; tail -1 gpurun_out/bench3.log | cut -c1-1500; cat gpurun_out/cpu3.log | tail -5
