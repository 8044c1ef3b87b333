#!/bin/bash
# Round 4, GPU call 1 (prepared at the end of round 3, when the GPU budget was spent): re-establish the baseline on the new box and
# collect what round 3 could not: SQ counters of the kernels written in its second half, and clean A/Bs of its last two switches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
# 1. the whole GPU suite (DCNv2 with one barrier per k-step is the default since the last minutes of round 3: its device parity ran on the
#    DCN checks only)
MQ_LADDER_OUT=$R/gpurun_out/r04c1_ladder.jsonl timeout 1200 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r04c1_pytest.log 2>&1; tail -8 gpurun_out/r04c1_pytest.log | cut -c1-300
# 2. end-to-end A/Bs, one switch per run, default twice (boxes differ by +-2 % run to run)
for v in NONE=0 MQ_FPN_TOPDOWN_FUSED=0 MQ_DCN_SYNC=2 MQ_SWIN_QKV_FUSED=1 MQ_SWIN_QKV_FUSED=0 MQ_VLFUSE_I2T_VARIANT=1 NONE=1; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-extras > gpurun_out/r04c1_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r04c1_ab_$v.log | cut -c1-140)"
done
# 3. SQ counters of the default step (two --pmc passes, --kernel-trace only): window_attn_qkv_kernel, swin_mlp2_tail_kernel, dyrelu_ln_kernel,
#    vlfuse_i2t_kernel<3,1,1,0,true>, dcn_igemm8_kernel<16,0,1> are new since profiles/r03_call4_sq_summary.txt
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$n -o b -- python $R/bench.py --steps 1 --warmup 2 --no-extras --no-graph > $R/gpurun_out/r04c1_pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|window_attn_qkv|swin_mlp2|dyrelu_ln|vlfuse_|dcn_igemm8|conv3x3_small2|align_fused" $f > $R/gpurun_out/r04c1_sq_$n.csv
done
cd $R
# 4. the default bench as the driver runs it
( time timeout 900 python bench.py ) > gpurun_out/r04c1_bench_default.log 2> gpurun_out/r04c1_bench_default.time; tail -1 gpurun_out/r04c1_bench_default.log | cut -c1-300
