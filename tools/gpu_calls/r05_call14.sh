#!/bin/bash
# Round 5, GPU call 14: the final commit once more on the kernels touched last (DCNv2 without the FENCE launch paths), and two runtime
# environment A/Bs: HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory), TORCH_BLAS_PREFER_HIPBLASLT=0 (rocBLAS for the library GEMMs).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "(test_block and (check_dcn or check_dyconv or check_swin_fpn or check_full_model)) or hip_graph_replay" > gpurun_out/r05c14_pytest.log 2>&1; tail -2 gpurun_out/r05c14_pytest.log | cut -c1-200
for i in 1 2; do
  for env in "MQ_NONE=0" "HIP_FORCE_DEV_KERNARG=1" "TORCH_BLAS_PREFER_HIPBLASLT=0"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c14_switch_ab.txt 2>&1; cat gpurun_out/r05c14_switch_ab.txt
