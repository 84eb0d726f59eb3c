#!/usr/bin/env python3
"""Disassembly of ONE kernel out of an object / shared library with embedded gfx950 code: dump_isa.py file substring [out]"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from check_isa import code_objects, OBJDUMP  # noqa: E402


def main():
    path, sub = sys.argv[1], sys.argv[2]
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "-C", f.name], capture_output=True, text=True, check=True).stdout
        keep, lines = False, []
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                keep = sub in m.group(1)
                if keep:
                    lines.append(line)
                continue
            if keep:
                lines.append(line)
        if lines:
            text = "\n".join(lines)
            if len(sys.argv) > 3:
                open(sys.argv[3], "w").write(text)
            else:
                print(text)
            return
    print("not found")


if __name__ == "__main__":
    main()
