#!/usr/bin/env python3
"""HBM traffic of the bench's kernel families from two rocprofv3 PMC passes (MI355X_MICROARCH.md, "HBM" and
"rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic_r01.json

Units and the gfx950 correction as the guide prescribes: FETCH_SIZE / WRITE_SIZE are reported in KiB
(TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced
reads at 64 B, i.e. HALF the bytes -- all loads of these kernels are 16 B/lane, so the read figure is doubled.
WRITE_SIZE is uncalibrated on this part and reported as is.  Steps are counted by the dispatches of
conv_small_n_kernel<1, *> (the vocoder's conv_post: exactly one per generation step).

    python tools/pmc_traffic.py <fetch dir> <write dir> distill <train steps in the profiled run>

does the same for `bench.py --mode distill` (per distillation step; the step count is given: warmup + steps + the
one profiled step; run the bench with --no-latency so that only training steps are in the trace)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name):
    if "resunit_kernel" in name or "conv_gemm_sk_kernel" in name or "ffn_geglu_kernel" in name:   # the fused vocoder ResBlock units, the stream-K launches, the fused feed-forward
        return "conv_gemm_kernel"                                     # (round 6) belong to the conv_gemm family
    if "adamw_ema2_zero_kernel" in name:                              # the one-pass optimizer tail (round 6)
        return "adamw_kernel"
    if "attention_plain2_kernel" in name:   # the self-attention forward kernel of round 3 (rounds 3-4 left it in "other": that
        return "attention_kernel"           # was the unexplained 35.6 GB of pmc_traffic_distill_r04.json)
    for key in ("conv_gemm_kernel", "conv1d_halo_kernel", "attention_kernel", "attn_bwd", "gn_", "layernorm", "ln_bwd", "geglu",
                "softmax_rows", "conv_small_n_kernel", "splitk_finish", "adamw_kernel", "ema2_kernel", "pack_weight",
                "wgrad_scatter", "wgrad_implicit", "wgrad_tn", "wgrad_rowsum", "im2col_t", "transpose", "adamw4_kernel",
                "add_slices", "linear_f32", "ctta_zero_kernel", "copy_segments", "concat", "pool2_sum", "zero_insert",
                "col_scatter", "heun_", "cfg_combine", "snr_mse", "nhwc", "time_features", "fourier_features", "gelu",
                "copyBuffer", "at::native"):
        if key in name:
            return {"adamw4_kernel": "adamw_kernel", "at::native": "torch_elementwise (at::native::*)"}.get(key, key.rstrip("_"))
    return "other"


OTHER = defaultdict(float)      # (counter, kernel name) -> KiB of the kernels no family claims: listed in the output


def load(directory, counter, step_kernel=r"conv_small_n_kernel<1[,>]"):
    """Sums `counter` by kernel family; a pass that left no CSV (the distillation-mode passes hang intermittently under
    --pmc and are killed by `timeout`) gives (None, 0).  Steps = dispatches of the once-per-step kernel `step_kernel`."""
    paths = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    if not paths:
        return None, 0
    tot, steps = defaultdict(float), 0
    seen = set()
    for row in csv.DictReader(open(paths[0])):
        if row["Counter_Name"] != counter:
            continue
        tot[family(row["Kernel_Name"])] += float(row["Counter_Value"])
        if family(row["Kernel_Name"]) == "other":
            OTHER[(counter, row["Kernel_Name"].split("(")[0][:70])] += float(row["Counter_Value"])
        if re.search(step_kernel, row["Kernel_Name"]) and row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            steps += 1
    return tot, max(1, steps)


def main():
    distill = len(sys.argv) > 3 and sys.argv[3] == "distill"
    # one micro-step = one launch of the loss-gradient kernel (every leg of `bench.py --mode distill` runs it once per micro-step)
    sk = r"snr_mse_grad_kernel" if distill else r"conv_small_n_kernel<1[,>]"
    fetch, fs = load(sys.argv[1], "FETCH_SIZE", sk)
    write, wsteps = load(sys.argv[2], "WRITE_SIZE", sk)
    unit = "GB per distillation micro-step (batch 9)" if distill else "GB per generation step (batch 32)"
    out = {"unit": unit, "steps_fetch_pass": fs, "steps_write_pass": wsteps,
           "fetch_correction": "x2 (gfx950: 128-B requests of 16 B/lane reads tallied at 64 B)", "families": {}}
    if fetch is None or write is None:
        out["missing_pass"] = "FETCH_SIZE" if fetch is None else "WRITE_SIZE"
    fams = set(fetch or {}) | set(write or {})
    key = lambda f: -((fetch or {}).get(f, 0) * 2 / max(fs, 1) + (write or {}).get(f, 0) / max(wsteps, 1))
    tr = tw = 0.0
    for fam in sorted(fams, key=key):
        rd = None if fetch is None else round(fetch.get(fam, 0.0) * 1024 * 2 / fs / 1e9, 3)
        wr = None if write is None else round(write.get(fam, 0.0) * 1024 / wsteps / 1e9, 3)
        tr += rd or 0.0
        tw += wr or 0.0
        out["families"][fam] = {"read_GB": rd, "write_GB": wr, "total_GB": None if rd is None or wr is None else round(rd + wr, 3)}
    top = sorted(OTHER.items(), key=lambda kv: -kv[1] * (2 if kv[0][0] == "FETCH_SIZE" else 1))[:8]
    out["other_top_kernels_GB_per_step"] = [{"counter": c, "kernel": k, "GB": round(v * 1024 * (2 if c == "FETCH_SIZE" else 1) / (fs if c == "FETCH_SIZE" else wsteps) / 1e9, 3)}
                                            for (c, k), v in top]
    out["all_kernels"] = {"read_GB": None if fetch is None else round(tr, 3), "write_GB": None if write is None else round(tw, 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
