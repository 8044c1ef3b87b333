#!/bin/bash
# Round 6, GPU call 23: where the torch-native tail of the eager step comes from (tools/tail_sites.py: aten operators by call site).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/tail_sites.py gpurun_out/r06c23_tail_sites.txt 2>&1 | tail -80
