"""One line per attention launch from two rocprofv3 --pmc passes (tools/pmc_attn.sh): cycles per XCD, MFMA-pipe busy
fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), VALU-issue busy = 4 x SQ_ACTIVE_INST_VALU /
(same), LDS busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs), waves per SIMD, and the three shares of wave time
(parked at s_waitcnt / s_barrier, issue-stalled, issuing).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles (MI355X_MICROARCH.md, cycle constants)."""
import csv, glob, sys, collections


def load(d):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "attn" not in r["Kernel_Name"]:
                continue
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], r["Grid_Size"])
            rows.setdefault(k, {})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(rows.items())


a, b = load(sys.argv[1]), load(sys.argv[2])
for (ka, va), (kb, vb) in zip(a, b):
    gui = vb.get("GRBM_GUI_ACTIVE", 0.0) / 8
    simd = max(gui * 1024, 1)
    wc = max(va.get("SQ_WAVE_CYCLES", 0.0), 1)
    name = ka[1]
    name = name[name.find("attn"):][:34]
    print(f"{name:34s} grid={ka[2]:>8s} cyc/XCD={gui:9.0f} MFMA={va.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)/simd:5.1%} "
          f"VALU={4*vb.get('SQ_ACTIVE_INST_VALU', 0)/simd:5.1%} LDS={vb.get('SQ_LDS_IDX_ACTIVE', 0)/max(gui*256,1):5.1%} "
          f"waves/SIMD={4*wc/simd:4.2f} parked={va.get('SQ_WAIT_ANY', 0)/wc:5.1%} stall={va.get('SQ_WAIT_INST_ANY', 0)/wc:5.1%} "
          f"(lds {va.get('SQ_WAIT_INST_LDS', 0)/wc:5.1%}) issuing={va.get('SQ_ACTIVE_INST_ANY', 0)/wc:5.1%} "
          f"valu_insts={va.get('SQ_INSTS_VALU', 0):.3e} lds_insts={vb.get('SQ_INSTS_LDS', 0):.3e} bank_conflict={va.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
