#!/bin/bash
# Round 6, GPU call 3: split-precise Swin MLP (weights packed already split, C = 384 through the main kernel with one W2 stage): (1) images/s at B = 8
# + per-kernel times; (2) the f32 tests that touch it; (3) the DEFAULT bench run as the driver issues it (wall time, line size, extras file).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python bench.py --dtype f32 --batch 8 --steps 5 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c3_bench_f32_b8_extras.json > gpurun_out/r06c3_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c3_bench_f32_b8.log | cut -c1-300
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r06c3_bench_f32_b8_extras.json"))
    for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:22]:
        print(f"{v:8.3f} ms  {k}")
except Exception as e:
    print("no extras:", e)
P
MQ_LADDER_OUT=$R/gpurun_out/r06c3_f32_ladder.jsonl timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "f32 and (swin or full_model or benchmark_configuration or mlp)" > gpurun_out/r06c3_pytest_f32.log 2>&1; tail -6 gpurun_out/r06c3_pytest_f32.log | cut -c1-500
/usr/bin/time -v timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06c3_bench_default.log 2> gpurun_out/r06c3_bench_default.err; tail -1 gpurun_out/r06c3_bench_default.log | wc -c; tail -1 gpurun_out/r06c3_bench_default.log; grep -E "Elapsed|Maximum resident" gpurun_out/r06c3_bench_default.err
cp bench_extras.json gpurun_out/r06c3_bench_extras.json 2>/dev/null
