"""MFMA-busy summary per kernel family from the rocprofv3 counter pass of scripts/pmc_collect.sh (pmc_<tag>_mfma.csv).
usage: pmc_mfma.py <csv> <out.json>"""
import collections
import csv
import json
import sys

FAMILIES = [("conv3x3_halo_persistent", "conv3x3_halo_pkernel"), ("conv3x3_halo", "conv3x3_halo_kernel"), ("conv_igemm2", "conv_igemm2_kernel"), ("conv_tap4_halo", "conv_tap4_halo_kernel"), ("conv_pw", "conv_pw_kernel"),
            ("wgrad3x3_halo", "wgrad3x3_halo_kernel"), ("conv_wgrad2", "conv_wgrad2_kernel"), ("wgrad_reduce", "wgrad_reduce_kernel"),
            ("bn_act_fwd", "bn_act_fwd"), ("bn_bwd_reduce", "bn_bwd_reduce"), ("bn_bwd_apply", "bn_bwd_apply")]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        for fam, pat in FAMILIES:
            if pat in r["Kernel_Name"]:
                agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[fam][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                break
    out = {"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE over "
                   "`bench.py --steps 2 --warmup 1` (scripts/pmc_collect.sh); per-launch averages per kernel family. mfma_busy_frac = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time (reads high on "
                   "launches < 0.3 ms, MI355X_MICROARCH.md)", "kernels": {}}
    for fam, _ in FAMILIES:
        if fam not in agg:
            continue
        n = len(disp[fam])
        us = sum(disp[fam].values()) / n
        k = {"launches": n, "avg_us": round(us, 2)}
        for c, v in sorted(agg[fam].items()):
            k[c] = round(v / n, 1)
        if k.get("SQ_BUSY_CU_CYCLES"):
            k["mfma_busy_frac"] = round(k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * k["SQ_BUSY_CU_CYCLES"]), 4)
        if k.get("SQ_WAVE_CYCLES"):
            k["wait_inst_any_frac_of_wave_cycles"] = round(k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"], 4)
        if k.get("GRBM_GUI_ACTIVE"):
            k["clock_ghz_from_gui_active"] = round(k["GRBM_GUI_ACTIVE"] / 8 / us * 1e-3, 3)
        out["kernels"][fam] = k
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for fam, k in out["kernels"].items():
        print(f"{fam:16s} {k['launches']:5d} x {k['avg_us']:8.2f} us  mfma_busy {k.get('mfma_busy_frac')}  clock {k.get('clock_ghz_from_gui_active')}")


if __name__ == "__main__":
    main()
