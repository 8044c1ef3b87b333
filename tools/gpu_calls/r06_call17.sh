#!/bin/bash
# Round 6, GPU call 17: per-tile fixed cost of the DCNv2 launch (fit over C = 128 .. 512), with LDS-copied and with register-staged weights.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c17_dcn_fixed_cost.json 2>&1 | tail -6
MQ_DCN_BDMA=0 timeout 300 python tools/dcn_fixed_cost_probe.py gpurun_out/r06c17_dcn_fixed_cost_regs.json 2>&1 | tail -6
