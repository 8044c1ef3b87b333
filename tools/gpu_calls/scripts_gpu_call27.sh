mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof27 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof27.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/prof27/*kernel_trace.csv.bak
tail -1 gpurun_out/bench_prof27.log | cut -c1-200
