# round 2, GPU call 4: Swin MLP tile variants, MQ-GLIP-L family after the LayerNorm width fix, fused-width A/B, LVIS protocol
# with chunk batching
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 2; do MQ_MICRO_ONLY=swin MQ_SWIN_MLP_VARIANT=$v timeout 120 python tools/microbench.py 2>&1 | grep kernel | cut -c1-200; done
( time timeout 900 python -m pytest tests -m gpu -q -k "glip_l or swin_mlp or caches or dyconv or dcn" ) > gpurun_out/r02_pytest4.log 2>&1; tail -5 gpurun_out/r02_pytest4.log; grep -E "^E  .*(max_err|Error)" gpurun_out/r02_pytest4.log | cut -c1-220 | head -20
for w in "96,192,384" "96,192" "96" ""; do MQ_SWIN_MLP_WIDTHS="$w" timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 > gpurun_out/r02_bench4_w.log 2>&1; echo "widths=[$w]"; grep -E '^\{' gpurun_out/r02_bench4_w.log | cut -c1-190; done
timeout 600 python bench.py --workload mq-glip-l --steps 5 > gpurun_out/r02_bench4_glipl.log 2>&1; grep -E '^\{' gpurun_out/r02_bench4_glipl.log | cut -c1-300; tail -2 gpurun_out/r02_bench4_glipl.log | cut -c1-200
timeout 400 python bench.py --workload lvis --batch 1 --steps 4 --warmup 2 --chunk-batch 31 > gpurun_out/r02_bench4_lvis_b1_cb31.log 2>&1; grep -E '^\{' gpurun_out/r02_bench4_lvis_b1_cb31.log | cut -c1-300; tail -2 gpurun_out/r02_bench4_lvis_b1_cb31.log | cut -c1-200
timeout 400 python bench.py --workload lvis --batch 8 --steps 2 --warmup 2 --chunk-batch 32 > gpurun_out/r02_bench4_lvis_b8_cb32.log 2>&1; grep -E '^\{' gpurun_out/r02_bench4_lvis_b8_cb32.log | cut -c1-300
