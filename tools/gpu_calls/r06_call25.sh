#!/bin/bash
# Round 6, GPU call 25: VLFuse kernels with strided key / value operands (ABI 30: views of the projection output, no layout copies): parity in the
# three builds incl. MQ-GroundingDINO, benches, per-step kernel statistics.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1800 python -m pytest tests/test_gpu_parity.py -q -x -k "check_vlfuse or check_vl_fuse or fusion_layer or check_full_model or benchmark_configuration or groundingdino or gdino or (f32_block and (vlfuse or vl_fuse or full_model)) or (bf16_block and (vlfuse or vl_fuse or full_model))" > gpurun_out/r06c25_pytest.log 2>&1; tail -3 gpurun_out/r06c25_pytest.log | cut -c1-300
for i in 1 2 3; do
  echo -n "fp16: "; timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/r06c25_bench.txt
echo -n "split: "; timeout 200 python bench.py --dtype f32 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/r06c25_bench.txt
cd /tmp
for n in 4 24; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o bench -- python $R/bench.py --steps $n --warmup 2 --no-graph --no-extras > /dev/null 2>&1
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r06c25_kernel_stats_steps$n.csv
done
cd $R
python tools/per_step_kernel_stats.py gpurun_out/r06c25_kernel_stats_steps4.csv 4 gpurun_out/r06c25_kernel_stats_steps24.csv 24 gpurun_out/r06c25_per_step_kernel_stats.csv | tee gpurun_out/r06c25_per_step_summary.txt
