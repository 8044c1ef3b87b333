#!/bin/bash
# Round 3, GPU call 1 (prepared at the end of round 2, when no GPU minutes were left): first run ON THE DEVICE of everything that was
# only checked through tests/simt -- the bf16 builds of all kernels, SCORE_AGG modes, the resident-key attention kernel -- then the A/Bs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. the whole GPU suite (new this call: test_bf16_*, check_score_agg, test_resident_attention_kernel)
MQ_GPU_FULL=1 timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r03_pytest1.log 2>&1; tail -6 gpurun_out/r03_pytest1.log | cut -c1-300
# 2. headline bench, then the same with the text-sized attentions on the resident-key kernel (lang_path_b64 carries the north-star number)
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experimental > gpurun_out/r03_bench1_default.log 2>&1; tail -1 gpurun_out/r03_bench1_default.log | cut -c1-200
MQ_ATTN_RESIDENT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench1_resident.log 2>&1; tail -1 gpurun_out/r03_bench1_resident.log | cut -c1-200
for f in default resident; do python - <<PY
import json
d = json.loads(open("gpurun_out/r03_bench1_$f.log").read().strip().splitlines()[-1])
lp = d.get("lang_path_b64", {})
print("$f", d["value"], {k: lp.get(k) for k in ("ms_language_path", "attention_kernels_ms", "attention_mfma_utilisation")}, lp.get("kernels_ms"))
PY
done
MQ_LN_VARIANT=2 MQ_OFFSET_CONV_VARIANT=2 MQ_PATCH_MERGE_FUSED=1 MQ_FPN_VIA_DCN=1 MQ_NMS_EARLY_STOP=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lang-b64 --no-experimental > gpurun_out/r03_bench1_ln2.log 2>&1; tail -1 gpurun_out/r03_bench1_ln2.log | cut -c1-200
# 3. BASELINE configs[3]: MQ-GLIP-L on bf16 MFMA (default of the workload) next to fp16
timeout 200 python bench.py --workload mq-glip-l --steps 10 --warmup 3 > gpurun_out/r03_bench1_glipl_bf16.log 2>&1; tail -1 gpurun_out/r03_bench1_glipl_bf16.log | cut -c1-200
timeout 200 python bench.py --workload mq-glip-l --dtype f16 --steps 10 --warmup 3 > gpurun_out/r03_bench1_glipl_f16.log 2>&1; tail -1 gpurun_out/r03_bench1_glipl_f16.log | cut -c1-200
# 4. MQ-GroundingDINO with the resident kernel (text enhancer 4 x 64, decoder text cross-attention 8 x 32)
MQ_ATTN_RESIDENT=1 timeout 200 python bench.py --workload mq-gdino-t --steps 10 --warmup 3 > gpurun_out/r03_bench1_gdino_resident.log 2>&1; tail -1 gpurun_out/r03_bench1_gdino_resident.log | cut -c1-200
# 5. library GEMMs are 5.3 ms of the 24.3 ms step: what does PyTorch's TunableOp (rocBLAS / hipBLASLt solution
#    search per GEMM shape, results kept in a CSV that later runs re-use) buy?  Bounded tuning time per shape.
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  PYTORCH_TUNABLEOP_FILENAME=gpurun_out/r03_tunableop.csv timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lang-b64 --no-experimental \
  > gpurun_out/r03_bench1_tunableop.log 2>&1; tail -1 gpurun_out/r03_bench1_tunableop.log | cut -c1-200; wc -l gpurun_out/r03_tunableop*.csv 2>/dev/null
# 6. the opt-in variants one by one (attribution; the default bench run of round 2 only A/B'd them as a group)
for v in MQ_LN_VARIANT=2 MQ_OFFSET_CONV_VARIANT=2 MQ_PATCH_MERGE_FUSED=1 MQ_FPN_VIA_DCN=1 MQ_NMS_EARLY_STOP=1; do
  env $v timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lang-b64 --no-experimental > gpurun_out/r03_bench1_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r03_bench1_$v.log | cut -c1-120)"
done
# 7. kernel statistics of the default step and of the step with every variant on (eager launches; only the stats CSVs are kept)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph \
  --no-cpu-baseline --no-lang-b64 --no-experimental > $R/gpurun_out/r03_prof_default.log 2>&1
MQ_LN_VARIANT=2 MQ_OFFSET_CONV_VARIANT=2 MQ_PATCH_MERGE_FUSED=1 MQ_FPN_VIA_DCN=1 MQ_NMS_EARLY_STOP=1 MQ_ATTN_RESIDENT=1 \
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_variants -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-graph \
  --no-cpu-baseline --no-lang-b64 --no-experimental > $R/gpurun_out/r03_prof_variants.log 2>&1
cd $R
for t in default variants; do f=$(find /tmp/prof_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_call1_${t}_kernel_stats.csv && head -25 $f | cut -c1-160; done

# 0. lane layouts the emulation assumes (32x32x16 MFMA is not used by a shipped kernel yet)
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/mfma_layout_probe.hip -o /tmp/mfma_probe 2>/dev/null && /tmp/mfma_probe
hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_microbench.hip -o /tmp/mfma_lds 2>/dev/null && /tmp/mfma_lds | tee gpurun_out/r03_mfma_lds_microbench.txt
