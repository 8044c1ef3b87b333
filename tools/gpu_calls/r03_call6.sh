#!/bin/bash
# Round 3, GPU call 6: where does a VLFuse image-side step wait (ablation timings), text side with the live row blocks packed,
# clean end-to-end A/Bs of the Swin MLP flags, the parity case that failed in call 5, kernel stats + default bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_MICRO_ONLY=vlfuse timeout 400 python tools/microbench.py gpurun_out/r03c6_micro_vlfuse.json > gpurun_out/r03c6_micro_vlfuse.log 2>&1; grep kernel gpurun_out/r03c6_micro_vlfuse.log | cut -c1-230
for v in NONE=0 MQ_SWIN_MLP2_FLAGS=3 MQ_SWIN_MLP2_FLAGS=2 MQ_SWIN_MLP2_FLAGS=0 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r03c6_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03c6_ab_$v.log | cut -c1-140)"
done
MQ_LADDER_OUT=$R/gpurun_out/r03c6_ladder.jsonl timeout 900 python -m pytest tests -q -m gpu -k "benchmark_configuration_parity or vlfuse or swin_mlp or test_block" > gpurun_out/r03c6_pytest.log 2>&1; tail -8 gpurun_out/r03c6_pytest.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r03c6_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r03c6_kernel_stats.csv && head -14 $f | cut -c1-170
cd $R
( time timeout 900 python bench.py ) > gpurun_out/r03c6_bench_default.log 2> gpurun_out/r03c6_bench_default.time; tail -1 gpurun_out/r03c6_bench_default.log | cut -c1-300; tail -4 gpurun_out/r03c6_bench_default.time
