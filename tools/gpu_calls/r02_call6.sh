# round 2, GPU call 6: PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate, kernel-trace only), fused-width A/B with 30 steps
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/pmc_traffic.sh > gpurun_out/r02_pmc6.log 2>&1; python tools/pmc_reduce.py gpurun_out/pmc_r02 gpurun_out/r02_pmc_traffic.json | tail -40
for w in "96,192,384" "96,192" ""; do MQ_SWIN_MLP_WIDTHS="$w" timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 --steps 30 --warmup 5 > gpurun_out/r02_bench6_w.log 2>&1; echo "widths=[$w]"; grep -E '^\{' gpurun_out/r02_bench6_w.log | cut -c1-190; done
