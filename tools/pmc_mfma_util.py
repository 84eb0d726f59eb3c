#!/usr/bin/env python3
"""Matrix-pipe utilisation of the MFMA kernel families from ONE rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d D -o p -- python3 bench.py --mode gen \
        --steps 2 --warmup 1 --no-cpu-baseline --no-latency
    python tools/pmc_mfma_util.py D > profiles/pmc_mfma_util_r01.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles the matrix pipe was busy (16 per v_mfma_f32_16x16x32_bf16:
profiles/pmc_r01.md).  utilisation = counter / (1024 SIMDs x kernel duration x 2.4 GHz); the duration is the dispatch's
own Start/End timestamp of the same pass.  The equivalent rate, utilisation x 2.5 PFLOP/s, is the EXECUTED (padding
included) MFMA rate and should agree with bench.py's executed_tflops_incl_padding, which is computed from launch
shapes and HIP-event times -- a hardware cross-check of the roofline object."""
import csv
import glob
import json
import sys
from collections import defaultdict

SIMDS, CLOCK_HZ, PEAK_TFLOPS = 256 * 4, 2.4e9, 2500.0


def family(name):
    if "resunit_kernel" in name:
        return "conv_gemm_kernel"
    if "attention_plain2_kernel" in name:      # the self-attention forward kernel (round 5: listed; rounds 1-4 left it out)
        return "attention_plain2_kernel"
    for key in ("conv_gemm_kernel", "conv1d_halo_kernel", "attention_kernel", "conv_small_n_kernel"):
        if key in name:
            return key
    return None


def main():
    path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    busy, ns, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if fam is None or r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        busy[fam] += float(r["Counter_Value"])
        ns[fam] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        cnt[fam] += 1
    out = {"counter": "SQ_VALU_MFMA_BUSY_CYCLES", "simds": SIMDS, "clock_GHz": CLOCK_HZ / 1e9, "families": {}}
    tb = tn = 0.0
    for fam in sorted(busy, key=lambda f: -ns[f]):
        u = busy[fam] / (SIMDS * ns[fam] * 1e-9 * CLOCK_HZ) if ns[fam] else 0.0
        out["families"][fam] = {"dispatches": cnt[fam], "kernel_ms_total": round(ns[fam] / 1e6, 3),
                                "mfma_busy_cycles": busy[fam], "matrix_pipe_utilisation": round(u, 4),
                                "executed_TFLOPs_equiv": round(u * PEAK_TFLOPS, 1)}
        tb += busy[fam]
        tn += ns[fam]
    u = tb / (SIMDS * tn * 1e-9 * CLOCK_HZ) if tn else 0.0
    out["all_mfma_kernels"] = {"matrix_pipe_utilisation": round(u, 4), "executed_TFLOPs_equiv": round(u * PEAK_TFLOPS, 1)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
