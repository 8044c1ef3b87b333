#!/bin/bash
# Round 6, GPU call 21: DCNv2 band-order tiles with the row table in LDS: parity (fp16, split-precise), benches, the fixed-cost fit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "check_dcn or check_dyconv or check_ref_pins or check_swin_fpn or check_vlfuse or (f32_block and (dcn or dyconv))" > gpurun_out/r06c21_pytest.log 2>&1; tail -3 gpurun_out/r06c21_pytest.log | cut -c1-300
for i in 1 2 3; do
  for env in "MQ_DCN_RASTER=0" "MQ_DCN_RASTER=1"; do
    echo -n "fp16 $env: "; env $env timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done 2>&1 | tee gpurun_out/r06c21_ab.txt
timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c21_dcn_fixed_cost.json 2>&1 | tail -5
MQ_DCN_RASTER=0 timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c21_dcn_fixed_cost_patches.json 2>&1 | tail -5
