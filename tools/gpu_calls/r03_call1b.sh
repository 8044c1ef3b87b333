#!/bin/bash
# Round 3, GPU call 1: (a) bf16 list on the device for the first time, (b) MFMA / LDS microbench, (c) per-switch A/B of the opt-in
# kernels (30 steps each), (d) SQ counters of the Swin MLP / attention / LayerNorm kernels (one PMC pass, --kernel-trace only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_GPU_FULL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" > gpurun_out/r03_pytest_bf16.log 2>&1; tail -15 gpurun_out/r03_pytest_bf16.log | cut -c1-300
hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_microbench.hip -o /tmp/mfma_lds 2>/dev/null && timeout 120 /tmp/mfma_lds | tee gpurun_out/r03_mfma_lds_microbench.txt
for v in NONE=0 MQ_LN_VARIANT=2 MQ_OFFSET_CONV_VARIANT=2 MQ_PATCH_MERGE_FUSED=1 MQ_FPN_VIA_DCN=1 MQ_NMS_EARLY_STOP=1 MQ_ATTN_RESIDENT=1 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-lang-b64 --no-experimental > gpurun_out/r03_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03_ab_$v.log | cut -c1-160)"
done
cd /tmp
OUT=$R/gpurun_out/pmc_sq; mkdir -p $OUT
MQ_MICRO_ONLY=swin timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/swin -o b -- python $R/tools/microbench.py > $OUT/swin.log 2>&1
f=$(find $OUT/swin -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/swin_sq.csv; grep -E "swin_mlp" $f >> $OUT/swin_sq.csv; }; rm -rf $OUT/swin
MQ_MICRO_ONLY=swin timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/swin2 -o b -- python $R/tools/microbench.py > $OUT/swin2.log 2>&1
f=$(find $OUT/swin2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/swin_sq2.csv; grep -E "swin_mlp" $f >> $OUT/swin_sq2.csv; }; rm -rf $OUT/swin2
tail -5 $OUT/swin.log; tail -3 $OUT/swin2.log
cd $R
MQ_ATTN_RESIDENT=1 MQ_LN_VARIANT=2 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/attn -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-experimental > $OUT/attn.log 2>&1
f=$(find $OUT/attn -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $OUT/attn_sq.csv; grep -E "attn_res|attn_chunk|attn_fwd|layernorm|swin_mlp|conv3x3_small|dcn_igemm|vlfuse" $f >> $OUT/attn_sq.csv; }; rm -rf $OUT/attn
ls -la $OUT
