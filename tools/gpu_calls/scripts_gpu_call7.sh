set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/gpu_diag.py gpurun_out/diag7.json > gpurun_out/diag7.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "graph or boundary" > gpurun_out/pytest7.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench7.log 2>&1
cd /tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc7_fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc7_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc7_write -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc7_write.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc7_fetch gpurun_out/pmc7_write
# keep only rows of our kernels to stay small
for d in pmc7_fetch pmc7_write; do for f in gpurun_out/$d/*counter_collection.csv; do head -1 $f > $f.small; grep -E "attn_fwd|window_attn|dcn_im2col|dyconv|dyrelu" $f >> $f.small; rm $f; done; rm -f gpurun_out/$d/*kernel_trace.csv; done
grep -c PASS gpurun_out/diag7.log; grep -E "FAIL|EXCEPTION" gpurun_out/diag7.log; tail -3 gpurun_out/pytest7.log; tail -1 gpurun_out/bench7.log | cut -c1-1700
