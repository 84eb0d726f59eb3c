#!/usr/bin/env python3
"""Scans the device code inside libctta_hip.so for instruction forms this build must not contain.

`v_pk_*_f32 ... op_sel:[...]` -- a packed fp32 operation whose LOW lane takes the HIGH dword of a source pair.  On the
MI355X boxes of this pool `v_pk_fma_f32 vD, vA, vB, vC op_sel:[0,1,0]` returns wrong low-lane results while waves of another
kernel issue MFMAs on the same CU (tools/pk_hazard.py: 11 904 wrong lanes in 20 launches beside ctta_conv_gemm, 701 568
beside a kernel that only issues v_mfma_f32_16x16x32_bf16, 0 beside an HBM-copy or a plain fp32-VALU kernel, 0 alone, 0
for the scalar twin, 0 for the plain and the op_sel_hi forms).  clang's SLP vectoriser emits that form (8 sites in the
round-2 build, all in the small fp32 MLP kernel); it is the reason two engine handles on two streams gave run-to-run
different results.  build.sh therefore compiles with -fno-slp-vectorize, and this scan (also run by tests/) keeps the
form from coming back through hand-written vector code.

    python tools/check_isa.py [path/to/libctta_hip.so]      exit status 1 if a forbidden form is present
    python tools/check_isa.py --regs lib.so [out.json]      registers and waves per SIMD of every kernel (occupancy table)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FORBIDDEN = re.compile(r"\bv_pk_\w*f32\b.*\bop_sel:\[")


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", data, i + len(MAGIC))
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx" in triple and size > 0:
                yield triple, data[i + o:i + o + size]
        pos = i + len(MAGIC)


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_registers(path):
    """{kernel symbol: (vgpr_count, vgpr_spill_count, agpr_count)} from the code objects' metadata notes.  A tile that sits
    exactly at an occupancy step (128 VGPRs = four waves per SIMD: two 8-wave workgroups per CU) loses a workgroup per CU with
    ONE more register -- round 4: an epilogue variant compiled into every tile took 256x128x32 from 128 to 129 and its
    fused-GEGLU launches from 796 to 557 TFLOP/s, unnoticed for most of the round (tests/test_host_cpu.py pins the table)."""
    regs = {}
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"\.agpr_count:\s+(\d+)(.*?)(?=\.agpr_count:|\Z)", out, re.S):
            blk = m.group(2)
            name = re.search(r"\.name:\s+(\S+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if name and vg:
                regs[name.group(1)] = (int(vg.group(1)), int(sp.group(1)) if sp else 0, int(m.group(1)))
    return regs


def waves_per_simd(vgpr, agpr=0):
    """gfx950: one 512-entry register file per SIMD lane shared by architectural and accumulation registers, allocated in
    blocks of 8; at most 8 waves per SIMD.  In gfx90a+ code-object metadata `.vgpr_count` is ALREADY the unified total
    (align4(architectural) + accumulation; e.g. `ln_bwd_kernel<4,false>`: vgpr 264 with agpr 8), so occupancy follows from
    it alone -- `agpr` is accepted for the table's sake and ignored (round 4 added it on top and under-counted kernels
    with accumulation registers by one step: ADVICE r4)."""
    return max(1, min(8, 512 // max(8, (vgpr + 7) // 8 * 8)))


def occupancy_table(path):
    """{kernel: [vgpr, agpr, spilled, waves per SIMD]} -- `python tools/check_isa.py --regs lib.so out.json` writes it;
    tests/golden/kernel_occupancy.json is the committed table the CPU suite compares the built library with."""
    return {k: [v[0], v[2], v[1], waves_per_simd(v[0], v[2])] for k, v in sorted(kernel_registers(path).items())}


def scan(path):
    hits, n_insn, n_obj = [], 0, 0
    for triple, blob in code_objects(path):
        n_obj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
                continue
            if "v_" in line or "s_" in line:
                n_insn += 1
            if FORBIDDEN.search(line):
                hits.append((kernel, line.strip()))
    return hits, n_insn, n_obj


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--regs":      # --regs lib.so [out.json]: the occupancy table
        import json
        table = occupancy_table(sys.argv[2])
        if len(sys.argv) > 3:
            json.dump(table, open(sys.argv[3], "w"), indent=0, sort_keys=True)
        for k, v in table.items():
            print("%-100s vgpr %3d agpr %3d spilled %2d waves/SIMD %d" % (k[:100], v[0], v[1], v[2], v[3]))
        sys.exit(0)
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "consistencytta_amd", "libctta_hip.so")
    hits, n_insn, n_obj = scan(lib)
    print("%s: %d code objects, %d instructions scanned, %d forbidden packed-fp32 op_sel forms" % (lib, n_obj, n_insn, len(hits)))
    for k, l in hits[:20]:
        print("  %s: %s" % (k[:60], l))
    sys.exit(1 if hits else 0)
