#!/bin/bash
# Round 6, GPU call 1: (1) the default bench line (must be < 4 KB and parse); (2) the SPLIT-PRECISE mode (fp32 operands as hi + lo through three
# fp16 MFMAs, csrc/common.h) on the device: the f32 tests of the GPU suite with every row to a ladder file; (3) its images/s at B = 8 with the
# per-kernel HIP-event times; (4) rocprofv3 kernel stats of the split-precise step (eager).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 200 python bench.py --dtype f32 --batch 8 --steps 5 --warmup 2 --no-extras --extras-file $R/gpurun_out/r06c1_bench_f32_b8_extras.json > gpurun_out/r06c1_bench_f32_b8.log 2>&1; tail -1 gpurun_out/r06c1_bench_f32_b8.log | cut -c1-600
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r06c1_bench_f32_b8_extras.json"))
    for k, v in sorted(d["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:25]:
        print(f"{v:8.3f} ms  {k}")
except Exception as e:
    print("no extras:", e)
P
MQ_LADDER_OUT=$R/gpurun_out/r06c1_f32_ladder.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "f32" > gpurun_out/r06c1_pytest_f32.log 2>&1; tail -15 gpurun_out/r06c1_pytest_f32.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -o bench -- python $R/bench.py --dtype f32 --batch 8 --steps 3 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06c1_prof.log 2>&1
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06c1_f32_b8_kernel_stats.csv && head -30 $f | cut -c1-170
cd $R
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06c1_bench_default.log 2>gpurun_out/r06c1_bench_default.err; tail -1 gpurun_out/r06c1_bench_default.log | wc -c; tail -1 gpurun_out/r06c1_bench_default.log
cp bench_extras.json gpurun_out/r06c1_bench_extras.json 2>/dev/null
