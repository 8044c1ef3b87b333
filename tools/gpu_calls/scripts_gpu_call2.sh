set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu2.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu2.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof2.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof2 | head -20
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench2.log 2>&1
tail -3 gpurun_out/pytest_gpu2.log; tail -2 gpurun_out/bench2.log
