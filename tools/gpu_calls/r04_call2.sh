#!/bin/bash
# Round 4, GPU call 2: the B = 8 graph-replayed parity case (new), and a kernel TIMELINE of the replayed step (start / end / queue of every
# kernel) to see what sits on the critical path of the 20 ms step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/*; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MQ_LADDER_OUT=$R/gpurun_out/r04c2_ladder.jsonl timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "b8_graph or benchmark_configuration_parity" > gpurun_out/r04c2_pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r04c2_pytest.log | tail -5 | cut -c1-600
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_graph -o t -- python $R/bench.py --steps 3 --warmup 3 --no-extras > $R/gpurun_out/r04c2_trace.log 2>&1
f=$(find /tmp/trace_graph -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" $R/gpurun_out/r04c2_timeline.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~1/4 of the launches (the timed, replayed steps) with short names
keep = rows[-len(rows) // 4:]
t0 = int(keep[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,end_us,queue,stream,kernel\n")
    for r in keep:
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:.1f},{(int(r['End_Timestamp']) - t0) / 1e3:.1f},{r.get('Queue_Id', '')},{r.get('Stream_Id', '')},{r['Kernel_Name'][:60]}\n")
print(len(rows), "kernels,", len(keep), "kept")
PY
cd $R; tail -1 gpurun_out/r04c2_trace.log | cut -c1-200
