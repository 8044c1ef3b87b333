#!/bin/bash
# Round 6, GPU call 10: host profile of the replayed benchmark step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
timeout 600 python tools/host_profile.py 200 > gpurun_out/r06c10_host_profile.txt 2>&1; grep -v "^$" gpurun_out/r06c10_host_profile.txt | head -60 | cut -c1-200
