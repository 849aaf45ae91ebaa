"""Merge rocprofv3 --pmc counter_collection CSVs of SQ counters (one pass per counter group) into one per-kernel table:
    kernel, launches, <counter averages per launch ...>, mfma_busy, issue_stall, parked
python scripts/pmc_sq_to_profile.py out.csv pass1_counter_collection.csv [pass2 ...]

Derived columns (MI355X: 256 CUs x 4 SIMDs; units per MI355X_MICROARCH.md, rocprofv3 PMC section):
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of the launch in which a SIMD's matrix pipe was busy,
                averaged over all SIMDs of the chip.  The MFMA counter counts cycles summed over the SIMDs (calibration: exactly 32.0 per
                v_mfma_f32_16x16x4_f32 in every fp32 kernel = its issue time, mfma_cyc_per_inst below); GRBM_GUI_ACTIVE comes out summed
                over the 8 XCDs (one GRBM each: 8 x the kernel's cycles; e.g. 846k for a 44 us launch at ~2.4 GHz), hence the / 8
  issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (waves stalled at issue: MFMA operand / pipe dependencies)
  parked      = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (waves parked in s_waitcnt / barriers)
  mfma_cyc_per_inst = SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA (32 for back-to-back v_mfma_f32_16x16x4_f32; ~16 for 16x16x32 bf16)
The kernels of a step are the rows; bench.py reads `mfma_busy` of the kernel its roofline_mfma object describes from the newest file."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if "g4d::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
rows = []
for k, cs in acc.items():
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    g = avg.get("GRBM_GUI_ACTIVE", 0.0)
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    d = {"mfma_busy": avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * g / 8.0) if g else "",
         "issue_stall": avg.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else "",
         "parked": avg.get("SQ_WAIT_ANY", 0.0) / wc if wc else "",
         "mfma_cyc_per_inst": avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / avg["SQ_INSTS_MFMA"] if avg.get("SQ_INSTS_MFMA") else ""}
    rows.append((g * n, k, n, avg, d))
rows.sort(key=lambda r: -r[0])
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + counters + ["mfma_busy", "issue_stall", "parked", "mfma_cyc_per_inst"])
    for _, k, n, avg, d in rows:
        w.writerow([k, n] + [round(avg[c], 1) if c in avg else "" for c in counters] +
                   [round(d[x], 4) if d[x] != "" else "" for x in ("mfma_busy", "issue_stall", "parked", "mfma_cyc_per_inst")])
