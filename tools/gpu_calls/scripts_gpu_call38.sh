mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/pmc_traffic.sh > gpurun_out/pmc38.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_vlfuse
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/SQ -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-graph > $OUT/SQ.log 2>&1
f=$(ls $OUT/SQ/*counter_collection.csv | head -1); head -1 $f > $OUT/SQ.csv; grep -E "vlfuse_|dcn_igemm8|window_attn|conv3x3_small" $f >> $OUT/SQ.csv; rm -rf $OUT/SQ
ls -la $OUT
