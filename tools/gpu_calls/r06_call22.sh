#!/bin/bash
# Round 6, GPU call 22: Swin MLP per width, erf against table GELU, on the kernels with counted waits (the choice per width dates from round 3).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_MICRO_ONLY=swin timeout 600 python tools/microbench.py gpurun_out/r06c22_microbench_swin_mlp.json 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['kernel'], r['ms'], r['frac_of_mfma_peak'])"
