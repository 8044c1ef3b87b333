#!/bin/bash
# Round 5, GPU call 13: the DCNv2 kernel with scheduling fences behind its barriers (MQ_DCN_FENCE=1: the blend of the next half step can no longer be
# hoisted above the barrier, its gather is waited for a whole step later: vmcnt(4) instead of vmcnt(0) in the staging group): parity, A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
MQ_DCN_FENCE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "(test_block and (check_dcn or check_dyconv or check_ref_pins or check_swin_fpn))" > gpurun_out/r05c13_pytest.log 2>&1; tail -2 gpurun_out/r05c13_pytest.log | cut -c1-200
for i in 1 2 3; do
  for env in "MQ_DCN_FENCE=0" "MQ_DCN_FENCE=1"; do
    echo -n "$env: "; env $env timeout 90 python bench.py --steps 60 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=[x for x in d['rooflines'] if x['kernel'].startswith('dcn_igemm8_kernel (')][0]; print(d['value'], d['ms_per_step'], 'dcn ms/step', r['ms_per_step'], 'frac', r['frac'])"
  done
done > gpurun_out/r05c13_switch_ab.txt 2>&1; cat gpurun_out/r05c13_switch_ab.txt
