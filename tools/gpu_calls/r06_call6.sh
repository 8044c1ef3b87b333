#!/bin/bash
# Round 6, GPU call 6: the new split-precise tests (B = 8 graph replay vs the oracle, MQ-GLIP-L family, DyConv after the epilogue's launch-bounds fix) and
# the split-precise rate; equivalence tests after the tolerance change.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r06c6_f32_ladder.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "f32 and (b8 or glip_l or dyconv or fusion_layer)" > gpurun_out/r06c6_pytest_f32.log 2>&1; tail -12 gpurun_out/r06c6_pytest_f32.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "caches or staggered or graph" > gpurun_out/r06c6_pytest_equiv.log 2>&1; tail -4 gpurun_out/r06c6_pytest_equiv.log | cut -c1-300
timeout 300 python bench.py --dtype f32 --batch 8 --steps 8 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c6_bench_f32_b8_extras.json > gpurun_out/r06c6_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c6_bench_f32_b8.log | cut -c1-1200
