#!/bin/bash
# GPU call 12: rocprofv3 kernel statistics of the MQ-GroundingDINO step, EAGER launches (--no-graph), small run, hard timeout
# (the graph-replayed B = 16 trace of call 8 never finished).  Only the stats CSV is kept (gpurun_out is capped at 64 MiB).
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof12 -o bench -- python ${GRAFT_REPO_ROOT}/bench.py --workload mq-gdino-t --batch 16 --steps 2 --warmup 2 --no-graph > ${GRAFT_REPO_ROOT}/gpurun_out/r02_bench12_prof.log 2>&1
echo "rocprof rc=$?"
cd ${GRAFT_REPO_ROOT}
f=$(find /tmp/prof12 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/r02_call12_gdino_b16_kernel_stats.csv; head -45 $f | cut -c1-180; fi
tail -1 gpurun_out/r02_bench12_prof.log | cut -c1-200
