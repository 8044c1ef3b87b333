#!/bin/bash
# Round 3, GPU call 15: the same A/B in the other order, twice (adjacent micro-benchmark runs differ by up to 5 % on these boxes).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
for s in 1 2 1 2; do echo "MQ_DCN_SYNC=$s: $(MQ_DCN_SYNC=$s MQ_MICRO_ONLY=dcn MQ_DCN_ABL_LIST=0 timeout 40 python tools/microbench.py 2>&1 | grep kernel | cut -c60-130)"; done
