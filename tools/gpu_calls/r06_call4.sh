#!/bin/bash
# Round 6, GPU call 4: (1) split-precise A/Bs: the fused text kernels / the fused Swin qkv attention against their unfused paths (fp32 library GEMM +
# spill-free attention kernels); (2) library GEMM probe (3 x fp16 with fp32 output against fp32); (3) replay-vs-eager bit equality under BLAS choices;
# (4) the DEFAULT bench run as the driver issues it (wall time, line size, extras).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for env in "MQ_NONE=0" "MQ_BERT_ATTN_QKV_FUSED=0" "MQ_GCP_ATTN_FUSED=0" "MQ_SWIN_QKV_FUSED=0" "MQ_SWIN_QKV_FUSED=1" "MQ_NONE=1"; do echo -n "$env: "; env $env timeout 200 python bench.py --dtype f32 --batch 8 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee gpurun_out/r06c4_split_ab.txt
timeout 300 python tools/gemm_out_dtype_probe.py 2>&1 | tee gpurun_out/r06c4_gemm_probe.txt | tail -12
for env in "MQ_X=0" "TORCH_BLAS_PREFER_HIPBLASLT=0" "MQ_PROBE_BLAS=hipblas" "MQ_PROBE_BLAS=hipblaslt"; do echo "== $env"; env $env timeout 300 python tools/replay_equality_probe.py 2>&1 | grep -E "SAME|DIFF|captured|Error|error" ; done | tee gpurun_out/r06c4_replay_equality.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06c4_bench_default.log 2> gpurun_out/r06c4_bench_default.err ) 2>&1 | grep real; tail -1 gpurun_out/r06c4_bench_default.log | wc -c; tail -1 gpurun_out/r06c4_bench_default.log
cp bench_extras.json gpurun_out/r06c4_bench_extras.json 2>/dev/null
