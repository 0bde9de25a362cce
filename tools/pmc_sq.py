"""Per-kernel averages of SQ counters from `rocprofv3 --pmc ... --kernel-trace --output-format csv` runs.

    python tools/pmc_sq.py <substring of the kernel name> <dir> [<dir> ...]
"""
import csv
import glob
import sys
from collections import defaultdict

key = sys.argv[1]
acc, cnt = defaultdict(float), defaultdict(int)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
avg = {k: acc[k] / cnt[k] for k in acc}
for k in sorted(avg):
    print(f"{k:28s} {avg[k]:16.0f}   ({cnt[k]} dispatches)")
w = avg.get("SQ_WAVES")
if w:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_MFMA"):
        if k in avg:
            print(f"{k} per wave: {avg[k] / w:.1f}")
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS"):
        if k in avg:
            print(f"{k} per wave: {4 * avg[k] / w:.0f} cycles")

if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
    # MfmaUtil of rocprofv3's derived metrics: MFMA-busy cycles summed over the SIMDs / (active cycles x SIMDs); GRBM_GUI_ACTIVE is
    # reported summed over the 8 XCDs (MI355X_MICROARCH.md), 1024 SIMDs on the chip
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    print(f"kernel active cycles (GRBM_GUI_ACTIVE / 8): {cyc:.0f};  MFMA busy / (cycles x 1024 SIMDs): {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in avg:
        print(f"MFMA F32 flops per launch (MOPS x 512): {avg['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512:.4g}")
