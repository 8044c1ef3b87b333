#!/bin/bash
# Round 4, GPU call 10: VLFuse text-side key split sweep at B = 4 / 8; window attention with the region test on border windows only
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "window_attention or swin_fpn" > gpurun_out/r04c10_pytest.log 2>&1; grep -E "passed|failed|Error|assert|Fatal" gpurun_out/r04c10_pytest.log | tail -4 | cut -c1-600
MQ_MICRO_ONLY=t2i_sweep timeout 200 python tools/microbench.py gpurun_out/r04c10_t2i_sweep.json > gpurun_out/r04c10_t2i_sweep.log 2>&1; grep kernel gpurun_out/r04c10_t2i_sweep.log | cut -c1-160
MQ_MICRO_ONLY=window timeout 200 python tools/microbench.py gpurun_out/r04c10_window.json > gpurun_out/r04c10_window.log 2>&1; grep kernel gpurun_out/r04c10_window.log | cut -c1-200
for v in "NONE=0" "NONE=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c10_ab_$v.log 2>&1
  echo "$v: rc=$? $(tail -1 gpurun_out/r04c10_ab_$v.log | cut -c1-200)"
done
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04c10_ab_NONE=1.log") if l.startswith("{")][-1])
print(d["kernels_ms_per_step"])
PY
