#!/bin/bash
# Round 4, GPU call 13: the whole GPU suite at HEAD with the ladder file (every row of every check) -- the source of tests/golden/device_measured.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r04c13_ladder.jsonl timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r04c13_pytest.log 2>&1; tail -16 gpurun_out/r04c13_pytest.log | cut -c1-300
timeout 200 python bench.py --workload mq-gdino-t --steps 5 --warmup 2 --no-extras > gpurun_out/r04c13_gdino.log 2>&1; echo "gdino: $(tail -1 gpurun_out/r04c13_gdino.log | cut -c1-400)"
