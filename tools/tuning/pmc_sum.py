"""Per-kernel summary of the two counter passes collected by pmc.sh (gpurun_out/NAME_pmc{1,2}.csv)."""
import csv, sys, collections
def load(f):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("conv_tile_kernel", "wgrad_tile", "conv1x1", "conv_slide")): continue
        n = n.replace("void (anonymous namespace)::", "").split("(")[0]
        key = (n, r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"])
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d
name = sys.argv[1]
d1, d2 = load(f"gpurun_out/{name}_pmc1.csv"), load(f"gpurun_out/{name}_pmc2.csv")
import os
d3 = load(f"gpurun_out/{name}_pmc3.csv") if os.path.exists(f"gpurun_out/{name}_pmc3.csv") else {}
m = lambda v: sum(v) / len(v)
for key in d1:
    a, b = d1[key], d2.get(key, {})
    wc = m(a["SQ_WAVE_CYCLES"])
    print(f"{key[0]} grid={key[1]} lds={key[2]} vgpr={key[3]} n={len(a['SQ_WAVE_CYCLES'])//1} dur={m(a['_dur'])/1e3:.1f}us")
    print("   wave-cycles %.3g  wait_any %.1f%%  wait_inst_any %.1f%%  active_inst_any %.1f%%  wait_inst_lds %.1f%%" % (
        wc, 100*m(a["SQ_WAIT_ANY"])/wc, 100*m(a["SQ_WAIT_INST_ANY"])/wc, 100*m(a["SQ_ACTIVE_INST_ANY"])/wc, 100*m(a["SQ_WAIT_INST_LDS"])/wc))
    busy = m(a["SQ_BUSY_CYCLES"])
    print("   busy_cycles %.3g  lds_idx_active %.3g  lds_bank_conflict %.3g (%.1f%% of active)" % (
        busy, m(a["SQ_LDS_IDX_ACTIVE"]), m(a["SQ_LDS_BANK_CONFLICT"]), 100*m(a["SQ_LDS_BANK_CONFLICT"])/max(m(a["SQ_LDS_IDX_ACTIVE"]),1)))
    # LDS_IDX_ACTIVE / BUSY_CYCLES: share of the SQ-busy cycles in which the LDS index pipe was working (both summed over the
    # chip's SQs, so the quotient is per CU); the guide's MI355X note: effective clock = GRBM_GUI_ACTIVE / 8 / wall time
    print("   lds_idx_active / busy_cycles %.2f" % (m(a["SQ_LDS_IDX_ACTIVE"]) / max(busy, 1)))
    c = d3.get(key)
    if c:
        print("   effective clock %.2f GHz (GRBM_GUI_ACTIVE / 8 / duration; reads high below 0.3 ms)" % (
            m(c["GRBM_GUI_ACTIVE"]) / 8 / m(c["_dur"])))
    if b:
        print("   mfma_busy %.3g  insts_mfma %.3g  active: lds %.3g vmem %.3g valu %.3g  insts_lds %.3g  lds_data_fifo_full %.3g cmd_fifo_full %.3g" % (
            m(b["SQ_VALU_MFMA_BUSY_CYCLES"]), m(b["SQ_INSTS_MFMA"]), m(b["SQ_ACTIVE_INST_LDS"]), m(b["SQ_ACTIVE_INST_VMEM"]),
            m(b["SQ_ACTIVE_INST_VALU"]), m(b["SQ_INSTS_LDS"]), m(b["SQ_LDS_DATA_FIFO_FULL"]), m(b["SQ_LDS_CMD_FIFO_FULL"])))
