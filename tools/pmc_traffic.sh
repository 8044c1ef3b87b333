#!/bin/bash
# HBM-side traffic of the hot hand-written kernels from rocprofv3 PMC counters (bench.py "roofline.traffic").
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots) -> two passes, each with --kernel-trace only
# (never combine --pmc with sys / hip / hsa traces on this pool).  Eager launches (--no-graph) so every kernel is a
# separate dispatch.  Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh ; then
#   python tools/pmc_reduce.py gpurun_out/pmc_r04 profiles/r04_pmc_traffic.json      (MQ_ROUND selects the label)
set -x
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_${MQ_ROUND:-r04}      # MQ_BENCH_ARGS: extra bench.py arguments (e.g. "--dtype f32": the split-precise step)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o b -- \
    python ${GRAFT_REPO_ROOT:-$OLDPWD}/bench.py --steps 1 --warmup 2 --no-extras --no-graph ${MQ_BENCH_ARGS:-} > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  head -1 $f > $OUT/$c.csv; grep -E "vlfuse_|dcn_igemm8|swin_mlp|dyconv_fuse|layernorm|dyrelu_ln|window_attn|align_fused|conv3x3_|attn_resident|attn_text|attn_chunked|bert_attn|gcp_|patch_embed|post_" $f >> $OUT/$c.csv      # keep only the rows we reduce
  rm -rf $OUT/$c
done
