set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/gpu_diag.py gpurun_out/diag5.json > gpurun_out/diag5.log 2>&1
echo "diag exit $?" >> gpurun_out/diag5.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "graph or boundary" > gpurun_out/pytest5.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench5.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof5.log 2>&1
cd $GRAFT_REPO_ROOT
grep -c PASS gpurun_out/diag5.log; grep -E "FAIL|EXCEPTION" gpurun_out/diag5.log; tail -3 gpurun_out/pytest5.log; tail -1 gpurun_out/bench5.log | cut -c1-1800
