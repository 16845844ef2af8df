"""Summarises the MFMA-utilisation PMC pass of tools/profile.sh into profiles/<tag>_c<ch>_pmc_mfma.csv.

    kernel cycles   = GRBM_GUI_ACTIVE / 8            (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back)
    MFMA pipe util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel cycles)
                      (the counter adds 32 cycles per v_mfma_f32_32x32x16_bf16 and 16 per 16x16x32, over all SIMDs)
    effective clock = kernel cycles / average kernel duration of the stats pass"""
import csv, glob, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, ch):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", f"pmc_mfma_c{ch}")
    f = max(glob.glob(os.path.join(src, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            a = acc.setdefault(r["Kernel_Name"], {})
            c = a.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    dur = {}
    st = os.path.join(ROOT, "profiles", f"{tag}_c{ch}_kernel_stats.csv")
    if os.path.exists(st):
        for r in csv.DictReader(open(st)):
            dur[r["Name"]] = float(r["AverageNs"])
    out = os.path.join(ROOT, "profiles", f"{tag}_c{ch}_pmc_mfma.csv")
    with open(out, "w") as o:
        o.write("kernel,launches,kernel_cycles(GRBM_GUI_ACTIVE/8),SQ_VALU_MFMA_BUSY_CYCLES,mfma_pipe_util,SQ_INSTS_VALU_MFMA_MOPS_BF16,"
                "SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_ACTIVE_INST_VALU,effective_clock_GHz\n")
        for k, a in sorted(acc.items()):
            if "GRBM_GUI_ACTIVE" not in a:
                continue
            avg = lambda n: a[n][0] / a[n][1] if n in a else 0.0
            cyc = avg("GRBM_GUI_ACTIVE") / 8.0
            busy = avg("SQ_VALU_MFMA_BUSY_CYCLES")
            util = busy / (1024.0 * cyc) if cyc > 0 else 0.0
            clk = cyc / dur[k] if k in dur and dur[k] > 0 else 0.0
            o.write('"%s",%d,%.0f,%.0f,%.4f,%.0f,%.0f,%.0f,%.0f,%.3f\n' % (k, a["GRBM_GUI_ACTIVE"][1], cyc, busy, util,
                    avg("SQ_INSTS_VALU_MFMA_MOPS_BF16"), avg("SQ_WAVE_CYCLES"), avg("SQ_WAIT_ANY"), avg("SQ_ACTIVE_INST_VALU"), clk))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", sys.argv[2] if len(sys.argv) > 2 else "128")
