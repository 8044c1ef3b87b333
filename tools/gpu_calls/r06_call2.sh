#!/bin/bash
# Round 6, GPU call 2: the split-precise DCNv2 / VLFuse kernels (planar hi / lo fp16 LDS tiles, split once at staging): (1) images/s at B = 8 with the
# per-kernel HIP-event times; (2) the f32 tests of the GPU suite incl. MQ-GroundingDINO in the split-precise mode, ROIAlign / extract_query, every row
# to a ladder file; (3) the RCCL tests of the detection / evaluator gathers; (4) rocprofv3 kernel stats of the split-precise step (eager).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python bench.py --dtype f32 --batch 8 --steps 5 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c2_bench_f32_b8_extras.json > gpurun_out/r06c2_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c2_bench_f32_b8.log | cut -c1-400
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r06c2_bench_f32_b8_extras.json"))
    for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:22]:
        print(f"{v:8.3f} ms  {k}")
except Exception as e:
    print("no extras:", e)
P
for env in "MQ_DCN_SYNC=2" "MQ_DCN_WAVES=16"; do echo -n "$env: "; env $env timeout 200 python bench.py --dtype f32 --batch 8 --steps 5 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "rccl" > gpurun_out/r06c2_pytest_rccl.log 2>&1; tail -5 gpurun_out/r06c2_pytest_rccl.log | cut -c1-600
MQ_LADDER_OUT=$R/gpurun_out/r06c2_f32_ladder.jsonl timeout 1800 python -m pytest tests/test_gpu_parity.py -q -k "f32" > gpurun_out/r06c2_pytest_f32.log 2>&1; tail -25 gpurun_out/r06c2_pytest_f32.log | cut -c1-500
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -o bench -- python $R/bench.py --dtype f32 --batch 8 --steps 3 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06c2_prof.log 2>&1
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06c2_f32_b8_kernel_stats.csv && head -16 $f | cut -c1-170
