#!/usr/bin/env python3
"""Lists kernels whose innermost loops wait for every memory load before asking for the next one.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -S csrc/backward.hip -o /tmp/backward.s
    python tools/isa_serial_loads.py /tmp/*.s

A loop body (label .. backward branch to it, no other label inside) with one or two `global_load` / `buffer_load`
instructions, an `s_waitcnt vmcnt(0)` and no MFMA is one memory round trip per iteration and lane: fine for a tail loop,
a latency chain when it is the main loop of a streaming kernel (round 4: the slab sums of `wgrad_scatter_rows_kernel` and
`splitk_finish_kernel`, the pixel loop of `gn_partial_kernel` -- LABNOTES.md 4e).  Prints `file kernel loads lines` per loop."""
import re
import sys


def scan(path):
    rows = []
    func, labels = None, {}
    lines = open(path).read().split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, labels = m.group(1), {}
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", line)
        if not (m and func and m.group(1) in labels):
            continue
        body = lines[labels[m.group(1)]:i]
        if any(re.match(r"^\.LBB", x) for x in body[1:]):
            continue                      # not innermost
        loads = sum(1 for x in body if re.search(r"\b(global_load|buffer_load)", x))
        serial = any("s_waitcnt vmcnt(0)" in x for x in body)
        if 1 <= loads <= 2 and serial and not any("mfma" in x for x in body):
            rows.append((path.split("/")[-1], func, loads, len(body)))
    return rows


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for r in scan(p):
            print("%-18s %-70s loads %d lines %d" % (r[0], r[1][:70], r[2], r[3]))
