#!/bin/bash
# Round 3, GPU call 14 (the last GPU minutes of the round): DCNv2 with ONE barrier per k-step (MQ_DCN_SYNC=1) against two.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
for s in 2 1; do MQ_DCN_SYNC=$s MQ_MICRO_ONLY=dcn MQ_DCN_ABL_LIST=0 timeout 60 python tools/microbench.py gpurun_out/r03c14_micro_dcn_sync$s.json 2>&1 | grep kernel | cut -c1-160; done
MQ_DCN_SYNC=1 timeout 60 python -m pytest tests -q -m gpu -k "check_dcn or check_dyconv or check_ref_pins" 2>&1 | tail -2 | cut -c1-200
MQ_DCN_SYNC=1 timeout 100 python bench.py --steps 30 --warmup 3 --no-extras 2>&1 | tail -1 | cut -c1-140
