"""Summarise two rocprofv3 --pmc passes over tools/pmc_gemm.py (pass 1: SQ wave / MFMA / wait counters, pass 2:
GRBM_GUI_ACTIVE + LDS counters) into one line per GEMM launch: MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), share of wave-cycles parked (SQ_WAIT_ANY) and issue-stalled (SQ_WAIT_INST_ANY).
usage: python tools/pmc_gemm_summary.py <dir1> <dir2> <label>"""
import csv, glob, sys, collections


def load(d):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
            rows.setdefault(k, {})[r["Counter_Name"]] = rows.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(rows.items())


a, b = load(sys.argv[1]), load(sys.argv[2])
for (ka, va), (kb, vb) in zip(a, b):
    gui = vb.get("GRBM_GUI_ACTIVE", 0.0)
    busy = va.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui / 8 * 1024, 1)
    wc = max(va.get("SQ_WAVE_CYCLES", 0.0), 1)
    print(f"{sys.argv[3]} {ka[1][-34:]:34s} grid={ka[2]:>8s} cycles/XCD={gui/8:10.0f} MFMA busy={busy:5.1%} "
          f"parked={va.get('SQ_WAIT_ANY', 0)/wc:5.1%} issue-stall={va.get('SQ_WAIT_INST_ANY', 0)/wc:5.1%} "
          f"lds-stall={va.get('SQ_WAIT_INST_LDS', 0)/wc:5.1%} bank-conflict={va.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
