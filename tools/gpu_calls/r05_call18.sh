#!/bin/bash
# round 5, call 18: the forward with the text weights in MFMA B-fragment order: whole-model parity + graph replay + the headline
mkdir -p gpurun_out/c18
export GPU_MAX_HW_QUEUES=8
timeout 330 python -m pytest tests/test_gpu_parity.py -q -x --durations=8 -k "test_full_model_without_vision_queries or test_benchmark_configuration_parity or test_hip_graph_replay_matches_eager or test_integration_md_operator_stubs_run_as_written or test_bert_layer or test_f32_full_model" 2>&1 | tail -14 | tee gpurun_out/c18/pytest.log
timeout 200 python bench.py --no-experimental --no-cpu-baseline > gpurun_out/c18/bench.json 2> gpurun_out/c18/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c18/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, json.dumps(d.get("lang_path_b64"))[:900])
PY
