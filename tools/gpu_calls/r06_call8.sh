#!/bin/bash
# Round 6, GPU call 7 / 8: split-precise Swin attention at one workgroup per CU (call 7), fragments split once (call 8): parity + rate.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "f32 and (window or swin or full_model)" > gpurun_out/r06c8_pytest_f32.log 2>&1; tail -4 gpurun_out/r06c8_pytest_f32.log | cut -c1-300
for i in 1 2; do timeout 300 python bench.py --dtype f32 --batch 8 --steps 8 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c8_bench_f32_b8_extras.json > gpurun_out/r06c8_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c8_bench_f32_b8.log | cut -c1-200; done
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r06c8_bench_f32_b8_extras.json"))
    for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:14]:
        print(f"{v:8.3f} ms  {k}")
except Exception as e:
    print("no extras:", e)
P
