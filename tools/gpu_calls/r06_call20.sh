#!/bin/bash
# Round 6, GPU call 20: (a) DCNv2 tiles of 128 consecutive positions in BAND order (8-row bands, column by column: patch-shaped footprints, no padded
# tiles) against 8 x 16 patches: parity in the three builds, A/B, fixed-cost fit; (b) VLFuse text side with LDS-copied tiles (MQ_VL_T2I_DMA=1): parity, A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "check_dcn or check_dyconv or check_ref_pins or check_swin_fpn or check_conv3x3 or check_full_model or fusion_layer or benchmark_configuration_parity or (f32_block and (dcn or dyconv or conv3x3 or full_model)) or (bf16_block and (dcn or dyconv or conv3x3 or full_model))" > gpurun_out/r06c20_pytest.log 2>&1; tail -3 gpurun_out/r06c20_pytest.log | cut -c1-300
MQ_VL_T2I_DMA=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "(check_vlfuse or check_vl_fuse or fusion_layer or check_full_model or benchmark_configuration_parity) and not f32" > gpurun_out/r06c20_pytest_t2i_dma.log 2>&1; tail -3 gpurun_out/r06c20_pytest_t2i_dma.log | cut -c1-300
for i in 1 2 3; do
  for env in "MQ_DCN_RASTER=0" "MQ_DCN_RASTER=1" "MQ_DCN_RASTER=1 MQ_VL_T2I_DMA=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
  for env in "MQ_DCN_RASTER=0" "MQ_DCN_RASTER=1"; do
    echo -n "split $env: "; env $env timeout 200 python bench.py --dtype f32 --steps 8 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done 2>&1 | tee gpurun_out/r06c20_ab.txt
timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c20_dcn_fixed_cost.json 2>&1 | tail -5
cd /tmp
for env in "MQ_VL_T2I_DMA=0" "MQ_VL_T2I_DMA=1"; do
  env $env timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$env -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-extras > /dev/null 2>&1
  f=$(find /tmp/prof_$env -name "*kernel_stats.csv" | head -1); echo "$env"; [ -n "$f" ] && grep -E "vlfuse|dcn_igemm8" $f | cut -c1-140
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r06c20_kernel_stats.txt
