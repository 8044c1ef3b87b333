# round 2, GPU call 2: ROIAlign / extract_query / fused Swin MLP parity, tightened benchmark-configuration tolerances,
# bench with and without the fused Swin MLP, kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r02_pytest2.log 2>&1; tail -12 gpurun_out/r02_pytest2.log; grep -E "^E  .*(max_err|Error)" gpurun_out/r02_pytest2.log | cut -c1-200 | head -30
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench2_default.log 2>&1; grep -E '^\{' gpurun_out/r02_bench2_default.log | cut -c1-300
MQ_SWIN_FUSED_MLP=0 timeout 300 python bench.py --no-cpu-baseline --no-lang-b64 > gpurun_out/r02_bench2_unfused.log 2>&1; grep -E '^\{' gpurun_out/r02_bench2_unfused.log | cut -c1-200
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-lang-b64 > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_prof2.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02_prof2/**/bench_kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    # keep the LAST graph replay only (one steady-state forward) as a compact timeline: name, start, duration, stream
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t_end = int(rows[-1]["End_Timestamp"])
    # the eager profiling pass comes last: take the window [t_end - 120 ms, ...] and cut at the largest gaps instead
    out = open("gpurun_out/r02_timeline2.csv", "w")
    out.write("name,start_us,dur_us,queue\n")
    t0 = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t_end - s > 400e6:
            continue
        t0 = t0 or s
        out.write(f"{r['Kernel_Name'][:60]},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{r.get('Queue_Id', '')}\n")
    out.close()
PY
find gpurun_out/r02_prof2 -name "*kernel_trace.csv" -delete; ls gpurun_out/r02_prof2 gpurun_out/r02_timeline2.csv
