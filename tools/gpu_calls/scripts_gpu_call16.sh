set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest16.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof16.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/pytest16.log; tail -1 gpurun_out/bench_prof16.log | cut -c1-300; ls gpurun_out/prof16
