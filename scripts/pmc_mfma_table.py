"""One table per matrix kernel instantiation from the rocprofv3 --pmc passes of scripts/pmc_mfma.sh.

    python scripts/pmc_mfma_table.py gpurun_out/pmc_<tag> > profiles/r04_pmc_mfma.md

Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over
SIMDs (32 per v_mfma_f32_16x16x4_f32); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
GRBM_GUI_ACTIVE = shader-clock cycles the GPU was busy during the dispatch, summed over the 8 XCDs (it comes out at 8 x
duration x clock: 19 "GHz" before the division below).
  mfma_busy   = MFMA_BUSY / (GUI_ACTIVE x 256 CUs x 4 SIMDs)       share of all matrix-pipe cycles that executed an MFMA
  tflops_pmc  = MOPS_F32 x 512 flop / duration                     (one MOPS unit = 512 flop: 16x16x4 MFMA = 2048 flop = 4 units)
  wait shares = SQ_WAIT_ANY (parked: s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
"""
import collections, csv, glob, os, re, sys

SIMDS = 256 * 4
XCDS = 8


def load(d):
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:90]
            val[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name][r["Counter_Name"]] += 1
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {n: ({c: val[n][c] / cnt[n][c] for c in val[n]}, sum(dur[n]) / len(dur[n]), len(dur[n])) for n in val}


def main(d):
    data = load(d)
    want = ("wino4_gemm_kernel", "conv_glds_kernel", "conv_mfma_kernel", "hc_iter_reg1_kernel", "assign_kernel", "conv_stem")
    rows = [(n, v) for n, v in data.items() if any(w in n for w in want)]
    rows.sort(key=lambda kv: -kv[1][1] * kv[1][2])
    print("| kernel | dispatches | avg us (profiled) | clock GHz | MFMA busy | TFLOP/s from MOPS | frac of 157.3 | active | issue stall | parked | LDS issue stall | LDS bank conflict / LDS active |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for n, (c, us, k) in rows:
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        ghz = gui / (us * 1e3) if us else 0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * SIMDS) if gui else 0
        tf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) * 512 / (us * 1e-6) / 1e12 if us else 0
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        wc_b = (c.get("SQ_WAIT_INST_ANY", 0) + c.get("SQ_WAIT_ANY", 0) + c.get("SQ_ACTIVE_INST_ANY", 0)) or 1
        print(f"| `{n}` | {k} | {us:.1f} | {ghz:.2f} | {busy:.3f} | {tf:.1f} | {tf / 157.3:.3f} | "
              f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc_b:.3f} | {c.get('SQ_WAIT_INST_ANY', 0) / wc_b:.3f} | {c.get('SQ_WAIT_ANY', 0) / wc_b:.3f} | "
              f"{c.get('SQ_WAIT_INST_LDS', 0) / wc_b:.3f} | {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_ACTIVE_INST_LDS', 0)):.3f} |")
    print()
    print("Counters per dispatch (averages): see pmc_mfma_raw.md next to this file.")


if __name__ == "__main__":
    main(sys.argv[1])
