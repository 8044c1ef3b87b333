#!/bin/bash
# Round 3, GPU call 4: (a) which kernel selection moves the 141-token error ladder (bisect, one oracle run), (b) SQ counters of swin_mlp2 (all
# flag variants, microbench) and of the default step's hot kernels.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python tools/gpu_calls/r03_bisect.py > gpurun_out/r03c4_bisect.log 2>&1; grep "^==" gpurun_out/r03c4_bisect.log | cut -c1-400; tail -3 gpurun_out/r03c4_bisect.log | cut -c1-300
cd /tmp
OUT=$R/gpurun_out/pmc; mkdir -p $OUT
MQ_MICRO_ONLY=swin timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/swin -o b -- python $R/tools/microbench.py > $OUT/swin.log 2>&1
f=$(find $OUT/swin -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/swin_sq.csv; grep -E "swin_mlp" $f >> $OUT/swin_sq.csv; }; rm -rf $OUT/swin
MQ_MICRO_ONLY=swin timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/swin2 -o b -- python $R/tools/microbench.py > $OUT/swin2.log 2>&1
f=$(find $OUT/swin2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/swin_sq2.csv; grep -E "swin_mlp" $f >> $OUT/swin_sq2.csv; }; rm -rf $OUT/swin2
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/step -o b -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-extras > $OUT/step.log 2>&1
f=$(find $OUT/step -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/step_sq.csv; grep -E "align_fused|attn_res|attn_chunk|layernorm|swin_mlp|conv3x3_small|dcn_igemm|vlfuse|window_attn|dyconv|dyrelu" $f >> $OUT/step_sq.csv; }; rm -rf $OUT/step
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/step2 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-extras > $OUT/step2.log 2>&1
f=$(find $OUT/step2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/step_sq2.csv; grep -E "align_fused|attn_res|attn_chunk|swin_mlp|dcn_igemm|vlfuse" $f >> $OUT/step_sq2.csv; }; rm -rf $OUT/step2
ls -la $OUT
