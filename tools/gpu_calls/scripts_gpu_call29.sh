mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" > gpurun_out/smoke29.log 2>&1; tail -2 gpurun_out/smoke29.log
( time timeout 600 python bench.py ) > gpurun_out/bench29_default.log 2>&1; grep -E '^\{|real' gpurun_out/bench29_default.log | cut -c1-250
timeout 300 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench29_b1.log 2>&1; tail -1 gpurun_out/bench29_b1.log | cut -c1-200
timeout 300 python bench.py --batch 3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench29_b3.log 2>&1; tail -1 gpurun_out/bench29_b3.log | cut -c1-200
timeout 300 python bench.py --batch 16 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench29_b16.log 2>&1; tail -1 gpurun_out/bench29_b16.log | cut -c1-200
