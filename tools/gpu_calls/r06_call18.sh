#!/bin/bash
# Round 6, GPU call 18: DCNv2 tiles of 128 consecutive output positions (raster) against 8 x 16 patches (MQ_DCN_RASTER=0): parity on the device in
# the three builds, A/B with 3 alternations (fp16 and split-precise), the per-tile fixed-cost fit again.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "check_dcn or check_dyconv or check_ref_pins or check_swin_fpn or check_conv3x3 or check_full_model or fusion_layer or benchmark_configuration or (f32_block and (dcn or dyconv or conv3x3 or full_model)) or (bf16_block and (dcn or dyconv or conv3x3 or full_model)) or groundingdino" > gpurun_out/r06c18_pytest.log 2>&1; tail -4 gpurun_out/r06c18_pytest.log | cut -c1-300
for i in 1 2 3; do
  for env in "MQ_DCN_RASTER=0" "MQ_DCN_RASTER=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
  for env in "MQ_DCN_RASTER=0" "MQ_DCN_RASTER=1"; do
    echo -n "split $env: "; env $env timeout 200 python bench.py --dtype f32 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done 2>&1 | tee gpurun_out/r06c18_dcn_raster_ab.txt
timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c18_dcn_fixed_cost.json 2>&1 | tail -5
