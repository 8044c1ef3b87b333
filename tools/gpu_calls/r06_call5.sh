#!/bin/bash
# Round 6, GPU call 5: (1) the self-equivalence tests with torch.equal (replay vs eager, cache hits, staggered lanes) + RCCL tests; (2) split-precise
# bench after the planar offset conv and the unfused text-kernel policy; (3) MQ-GroundingDINO in the split-precise mode (B = 16); (4) default run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "graph or caches or staggered or rccl or boundary" > gpurun_out/r06c5_pytest_equiv.log 2>&1; tail -12 gpurun_out/r06c5_pytest_equiv.log | cut -c1-400
timeout 300 python bench.py --dtype f32 --batch 8 --steps 8 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c5_bench_f32_b8_extras.json > gpurun_out/r06c5_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c5_bench_f32_b8.log | cut -c1-260
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r06c5_bench_f32_b8_extras.json"))
    for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:16]:
        print(f"{v:8.3f} ms  {k}")
except Exception as e:
    print("no extras:", e)
P
timeout 300 python bench.py --workload mq-gdino-t --dtype f32 --steps 3 --warmup 2 --no-extras > gpurun_out/r06c5_bench_gdino_f32.log 2>&1; tail -1 gpurun_out/r06c5_bench_gdino_f32.log | cut -c1-300
timeout 300 python bench.py --workload mq-gdino-t --steps 5 --warmup 2 --no-extras > gpurun_out/r06c5_bench_gdino_f16.log 2>&1; tail -1 gpurun_out/r06c5_bench_gdino_f16.log | cut -c1-300
MQ_LADDER_OUT=$R/gpurun_out/r06c5_f32_ladder.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "f32 and (dyconv or full_model or benchmark_configuration or bert or gcp)" > gpurun_out/r06c5_pytest_f32.log 2>&1; tail -4 gpurun_out/r06c5_pytest_f32.log | cut -c1-400
( time timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06c5_bench_default.log 2> gpurun_out/r06c5_bench_default.err ) 2>&1 | grep real; tail -1 gpurun_out/r06c5_bench_default.log | wc -c; tail -1 gpurun_out/r06c5_bench_default.log | cut -c1-3000
cp bench_extras.json gpurun_out/r06c5_bench_extras.json 2>/dev/null
