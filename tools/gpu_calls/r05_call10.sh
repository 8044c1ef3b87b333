#!/bin/bash
# Round 5, GPU call 10 (final state): rocprofv3 kernel stats + the two PMC traffic passes + the two SQ counter passes of the default step (eager), then
# the default bench exactly as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r05c10_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05c10_kernel_stats.csv && head -6 $f | cut -c1-150
cd $R
MQ_ROUND=r05 timeout 500 bash tools/pmc_traffic.sh > gpurun_out/r05c10_pmc.log 2>&1
python tools/pmc_reduce.py gpurun_out/pmc_r05 gpurun_out/r05_pmc_traffic.json > /dev/null 2>&1; ls gpurun_out/pmc_r05 | head -3
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$n -o b -- python $R/bench.py --steps 1 --warmup 2 --no-extras --no-graph > $R/gpurun_out/r05c10_pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|window_attn|swin_mlp2|dyrelu_ln|vlfuse_|dcn_igemm8|conv3x3_|align_fused|attn_text|attn_chunked|bert_attn|gcp_attn|layernorm2|patch_embed|post_|dyconv_" $f > $R/gpurun_out/r05c10_sq_$n.csv
done
cd $R
python tools/sq_reduce.py gpurun_out/r05c10_sq_SQ_WAVE_CYCLES.csv gpurun_out/r05c10_sq_SQ_INSTS_VALU.csv > gpurun_out/r05c10_sq_summary.txt 2>&1; head -5 gpurun_out/r05c10_sq_summary.txt | cut -c1-200
( time timeout 900 python bench.py ) > gpurun_out/r05c10_bench_default.log 2> gpurun_out/r05c10_bench_default.time; tail -1 gpurun_out/r05c10_bench_default.log | cut -c1-300; tail -3 gpurun_out/r05c10_bench_default.time
