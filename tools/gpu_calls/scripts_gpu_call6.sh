mkdir -p gpurun_out
timeout 600 python tests/determinism_diag.py > gpurun_out/determinism6.log 2>&1
grep -E "SAME|DIFF|Error|error" gpurun_out/determinism6.log
