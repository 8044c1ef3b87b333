#!/bin/bash
# Round 6, evidence at HEAD (a): rocprofv3 kernel stats of the default (fp16) step and of the split-precise step (eager), the two PMC traffic passes for
# both, the two SQ counter passes of the default step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06f_prof.log 2>&1
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06_final_bench_kernel_stats.csv && head -6 $f | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_split -o bench -- python $R/bench.py --dtype f32 --steps 5 --warmup 2 --no-graph --no-extras > $R/gpurun_out/r06f_prof_split.log 2>&1
f=$(find /tmp/prof_split -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r06_final_split_kernel_stats.csv && head -6 $f | cut -c1-150
cd $R
MQ_ROUND=r06 timeout 500 bash tools/pmc_traffic.sh > gpurun_out/r06f_pmc.log 2>&1
python tools/pmc_reduce.py gpurun_out/pmc_r06 gpurun_out/r06_pmc_traffic.json > /dev/null 2>&1; ls gpurun_out/pmc_r06 | head -3; head -c 600 gpurun_out/r06_pmc_traffic.json
MQ_ROUND=r06split MQ_BENCH_ARGS="--dtype f32" timeout 500 bash tools/pmc_traffic.sh > gpurun_out/r06f_pmc_split.log 2>&1
python tools/pmc_reduce.py gpurun_out/pmc_r06split gpurun_out/r06_pmc_traffic_split.json > /dev/null 2>&1; head -c 600 gpurun_out/r06_pmc_traffic_split.json
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$n -o b -- python $R/bench.py --steps 1 --warmup 2 --no-extras --no-graph > $R/gpurun_out/r06f_pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|window_attn|swin_mlp2|dyrelu_ln|vlfuse_|dcn_igemm8|conv3x3_|align_fused|attn_text|attn_chunked|bert_attn|gcp_attn|layernorm2|patch_embed|post_|dyconv_" $f > $R/gpurun_out/r06f_sq_$n.csv
done
cd $R
python tools/sq_reduce.py gpurun_out/r06f_sq_SQ_WAVE_CYCLES.csv gpurun_out/r06f_sq_SQ_INSTS_VALU.csv > gpurun_out/r06_final_sq_summary.txt 2>&1; head -5 gpurun_out/r06_final_sq_summary.txt | cut -c1-200
rm -f gpurun_out/r06f_sq_*.csv; rm -rf gpurun_out/pmc_r06 gpurun_out/pmc_r06split
