#!/usr/bin/env python3
"""Kernels of the built library by their number of `s_waitcnt vmcnt(0)` instructions (with instruction and store counts).

    python tools/isa_vmcnt0_scan.py [lib.so] [top]
    python tools/isa_vmcnt0_scan.py --json lib.so tests/golden/kernel_vmcnt0.json      (the table the CPU suite compares with)

A straight-line epilogue whose global stores sit inside divergent `if (row < M)` blocks makes the compiler wait `vmcnt(0)`
in front of every later memory-dependent instruction: every row sweep then waits for the previous store to be acknowledged
(round 2: the bf16 wide-store epilogue of conv_gemm; round 5: its fp32 twin -- found with this count).  Many `vmcnt(0)` next
to many stores is the signature; dependent load -> load chains (index maps) show up here too and are a different matter."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import code_objects, OBJDUMP  # noqa: E402


def scan(lib):
    """[(vmcnt(0) waits, instructions, global / buffer stores, demangled kernel name)] of every kernel in the library"""
    rows = []
    for _, blob in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "-C", f.name], capture_output=True, text=True).stdout
        name, waits, insns, stores = None, 0, 0, 0
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                if name:
                    rows.append((waits, insns, stores, name))
                name, waits, insns, stores = m.group(1), 0, 0, 0
                continue
            if not name:
                continue
            insns += 1
            if "s_waitcnt vmcnt(0)" in line:
                waits += 1
            if "_store_" in line and ("global" in line or "buffer" in line):
                stores += 1
        if name:
            rows.append((waits, insns, stores, name))
    return rows


def main():
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "consistencytta_amd", "libctta_hip.so")
    args = [a for a in sys.argv[1:] if a != "--json"]
    if args:
        lib = args[0]
    top = int(args[1]) if len(args) > 1 and args[1].isdigit() else 40
    rows = scan(lib)
    if "--json" in sys.argv:      # --json lib.so out.json: {kernel: waits} for tests/golden/kernel_vmcnt0.json
        import json
        json.dump({nm: w for w, _, _, nm in rows}, open(args[-1], "w"), indent=0, sort_keys=True)
        return
    rows.sort(reverse=True)
    for w, n, st, nm in rows[:top]:
        print("%4d vmcnt(0)  %6d instructions  %4d stores  %s" % (w, n, st, nm[:120]))


if __name__ == "__main__":
    main()
