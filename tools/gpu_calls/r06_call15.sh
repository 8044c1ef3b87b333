#!/bin/bash
# Round 6, GPU call 15: Swin MLP with the GELU table read consumed one stage later (no lgkmcnt(0) in the loop) and the window-attention projections
# with their weight fragments through a two-deep ring: parity on the device in the three builds, kernel stats, benches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "check_swin or check_window or check_full_model or benchmark_configuration_parity or (f32_block and (swin or window)) or (bf16_block and (swin or window))" > gpurun_out/r06c15_pytest.log 2>&1; tail -4 gpurun_out/r06c15_pytest.log | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06c15_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06c15_bench_kernel_stats.csv && grep -E "swin_mlp2|window_attn|dcn_igemm8" $f | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_split -o bench -- python $R/bench.py --dtype f32 --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06c15_prof_split.log 2>&1
f=$(find /tmp/prof_split -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06c15_split_kernel_stats.csv && grep -E "swin_mlp2|window_attn|dcn_igemm8" $f | cut -c1-160
cd $R
for i in 1 2 3; do
  echo -n "fp16: "; timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  echo -n "split: "; timeout 200 python bench.py --dtype f32 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done 2>&1 | tee gpurun_out/r06c15_bench.txt
