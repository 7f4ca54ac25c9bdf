"""HBM bytes per launch of the main kernel families from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv out.json

Units / corrections as MI355X_MICROARCH.md (HBM, rocprofv3): both counters are in KiB; gfx950 reports half the bytes of
wide coalesced reads, so FETCH_SIZE is doubled."""
import collections
import csv
import json
import sys

FAMILIES = [("conv_tap4_halo<bf16>", "conv_tap4_halo_kernel"), ("conv_pw<bf16>", "conv_pw_kernel"), ("conv3x3_halo<bf16>", "conv3x3_halo_"), ("conv_igemm2<bf16>", "conv_igemm2_kernelIDF16b"), ("wgrad3x3_halo<bf16>", "wgrad3x3_halo_kernel"),
            ("wgrad_reduce", "wgrad_reduce_kernel"), ("splitk_epilogue<bf16>", "splitk_epilogue_kernelIDF16b"), ("conv_wgrad2<bf16>", "conv_wgrad2_kernelIDF16b"),
            ("bn_act_fwd<bf16>", "bn_act_fwd_slots_kernelIDF16b"), ("bn_bwd_reduce<bf16>", "bn_bwd_reduce_kernelIDF16b"),
            ("bn_bwd_apply<bf16>", "bn_bwd_apply_slots_kernelIDF16b"), ("adamw", "adamw_kernel")]


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, pat in FAMILIES:
            if pat in r["Kernel_Name"]:
                a = agg[fam]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                break
    return agg


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1`; "
                   "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); KiB -> bytes x1024; "
                   "averages over every launch of the kernel family in the run", "kernels": {}}
    for fam, _ in FAMILIES:
        if fam in fetch and fam in write:
            rd = fetch[fam][0] / fetch[fam][1] * 1024 * 2
            wr = write[fam][0] / write[fam][1] * 1024
            out["kernels"][fam] = {"launches": fetch[fam][1], "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
                                   "hbm_bytes_per_launch": int(rd + wr)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
