#!/bin/bash
# Round 6, GPU call 9: does PyTorch TunableOp (per-shape selection among the hipBLASLt / rocBLAS solutions, recorded once, replayed from a file) buy the
# library GEMMs of the step anything?  baseline / tune (writes the file) / replay from the file, same box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 40 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo -n "baseline: "; timeout 200 bash -c "$(declare -f run); run"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$R/gpurun_out/tunableop_mi355x.csv PYTORCH_TUNABLEOP_VERBOSE=0
echo -n "tuning run: "; ( time PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 timeout 1500 bash -c "$(declare -f run); run" ) 2>&1 | grep -E "^[0-9]|real"
ls -la gpurun_out/; wc -l gpurun_out/tunableop_mi355x*.csv 2>/dev/null | tail -2
for i in 1 2; do echo -n "replay from file: "; PYTORCH_TUNABLEOP_TUNING=0 timeout 300 bash -c "$(declare -f run); run"; done
unset PYTORCH_TUNABLEOP_ENABLED
echo -n "baseline again: "; timeout 200 bash -c "$(declare -f run); run"
