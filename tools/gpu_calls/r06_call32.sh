#!/bin/bash
# Round 6, GPU call 32 (the last): the default path at HEAD after the pooled-token kernel became the default -- smoke(), the graph / cache / full-model /
# benchmark-configuration tests of the suite, one bench run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c32_smoke.log 2>&1; tail -1 gpurun_out/r06c32_smoke.log
timeout 420 python -m pytest tests/test_gpu_parity.py -q -x -k "(graph_replay_matches_eager or backbone_and_caption_caches or check_full_model or benchmark_configuration_parity or pooled_tokens or extract_query) and not glip_l" > gpurun_out/r06c32_pytest.log 2>&1; tail -2 gpurun_out/r06c32_pytest.log | cut -c1-200
echo -n "fp16: "; timeout 120 python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee gpurun_out/r06c32_bench.txt
