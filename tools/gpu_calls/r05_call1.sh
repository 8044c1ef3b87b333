#!/bin/bash
# Round 5, GPU call 1: (1) the PRECISE MODE on the device -- the f32 tests of the GPU suite (every kernel with fp32 operands, the tiny model, one
# fusion layer and the full-depth benchmark configuration at 1e-3 / zero elements outside atol = rtol = 1e-3), every row to a ladder file;
# (2) the open A/Bs of round 4 (grouped DyConv epilogue, clamps inside the BERT-copy kernels), 3 alternations x 60 steps; (3) rocprofv3
# kernel-trace stats at HEAD; (4) the precise mode's images/s.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r05c1_f32_ladder.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "f32" > gpurun_out/r05c1_pytest_f32.log 2>&1; tail -25 gpurun_out/r05c1_pytest_f32.log | cut -c1-400
for i in 1 2 3; do
  for env in "MQ_NONE=0" "MQ_DYCONV_EPILOGUE_GROUPED=1" "MQ_BERT_CLAMP_FUSED=1"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05c1_switch_ab.txt 2>&1; cat gpurun_out/r05c1_switch_ab.txt
timeout 200 python bench.py --dtype f32 --batch 2 --steps 3 --warmup 2 --no-extras > gpurun_out/r05c1_bench_f32.log 2>&1; tail -1 gpurun_out/r05c1_bench_f32.log | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r05c1_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05c1_kernel_stats.csv && head -12 $f | cut -c1-150
