mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest34.log 2>&1; tail -3 gpurun_out/pytest34.log
( time timeout 600 python bench.py ) > gpurun_out/bench34_default.log 2>&1; grep -E '^\{|real' gpurun_out/bench34_default.log | cut -c1-220
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench34_torchrun.log 2>&1; grep -E '^\{' gpurun_out/bench34_torchrun.log | cut -c1-200; tail -2 gpurun_out/bench34_torchrun.log | cut -c1-200
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof34 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof34.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof34/bench_kernel_trace.csv; ls gpurun_out/prof34
