"""Reduce the two SQ counter passes of a GPU call (rocprofv3 --pmc ..., --kernel-trace only; tools/gpu_calls/r04_call1.sh) to a per-kernel
summary: issue / stall / wait shares of the wave cycles, cycles per wave, LDS bank-conflict share, instructions per wave by class.

    python tools/sq_reduce.py gpurun_out/r04c1_sq_SQ_WAVE_CYCLES.csv gpurun_out/r04c1_sq_SQ_INSTS_VALU.csv > profiles/r04_call1_sq_summary.txt"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
meta = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        meta[k] = (int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]))
        agg[k]["_grid"] += float(r["Grid_Size"])
        cnt[k]["_grid"] += 1
print("rocprofv3 --pmc (two passes of 8 SQ counters, --kernel-trace only) on bench.py --no-graph --steps 1 --no-extras (default kernel selection), mean per launch.")
print("issue / stall / wait = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES; cyc/wave = 4 x SQ_WAVE_CYCLES / waves of the grid;")
print("lds_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; per wave = SQ_INSTS_* / SQ_WAVES.\n")
for k in sorted(agg):
    a, c = agg[k], cnt[k]
    m = lambda n: a[n] / c[n] if c.get(n) else 0.0      # noqa: E731
    wc = m("SQ_WAVE_CYCLES")
    if not wc:
        continue
    wg, vg, ag, lds = meta[k]
    waves = m("_grid") / 64.0
    line = (f"{k:64s} n={c['SQ_WAVE_CYCLES']:4d} wg={wg:5d} vgpr={vg:3d}+{ag:3d} lds={lds // 1024:3d}K | issue {100 * m('SQ_ACTIVE_INST_ANY') / wc:3.0f}% "
            f"stall {100 * m('SQ_WAIT_INST_ANY') / wc:3.0f}% wait {100 * m('SQ_WAIT_ANY') / wc:3.0f}% | cyc/wave {4 * wc / max(waves, 1):9.0f} "
            f"mfma_busy {m('SQ_VALU_MFMA_BUSY_CYCLES'):.2e} lds_conf {100 * m('SQ_LDS_BANK_CONFLICT') / max(m('SQ_LDS_IDX_ACTIVE'), 1):3.0f}%")
    w = m("SQ_WAVES")
    if w:
        line += " | per wave: " + " ".join(f"{n.lower()} {m('SQ_INSTS_' + n) / w:7.0f}" for n in ("VALU", "MFMA", "LDS", "SALU", "VMEM_RD", "VMEM_WR", "SMEM"))
    print(line)
