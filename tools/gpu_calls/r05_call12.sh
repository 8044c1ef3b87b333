#!/bin/bash
# Round 5, GPU call 12 (FINAL STATE): the GPU suite as the driver runs it (every row to a ladder file), smoke() as the driver runs it, and the default
# bench exactly as the driver runs it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
( time MQ_LADDER_OUT=$R/gpurun_out/r05c12_ladder.jsonl timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r05c12_pytest.log 2>&1; tail -6 gpurun_out/r05c12_pytest.log | cut -c1-300
grep -o '"name": "[^"]*of the batch vs the B = 1 forward[^"]*"' gpurun_out/r05c12_ladder.jsonl | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05c12_smoke.log 2>&1; tail -2 gpurun_out/r05c12_smoke.log | cut -c1-200
( time timeout 900 python bench.py ) > gpurun_out/r05c12_bench_default.log 2> gpurun_out/r05c12_bench_default.time; tail -1 gpurun_out/r05c12_bench_default.log | cut -c1-300; tail -3 gpurun_out/r05c12_bench_default.time
