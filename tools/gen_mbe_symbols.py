#!/usr/bin/env python3
"""Lists every mbelib-neo function the reference's own sources call: each `mbe_*(` in /root/reference/src + include that the
reference does not define itself (dsd-neo writes a definition's name at column 0).  Output: tests/golden/mbe_symbols_called.json,
the list tests/test_cabi_exports.py holds the library's exports against.  Run in the build container only (the reference tree is
not on the GPU box; the fixture is data: names + the first call site of each)."""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mbe_symbols_called.json")


def main():
    calls, defined, types = {}, set(), set()
    for top in ("src", "include"):
        for d, _, files in os.walk(os.path.join(REF, top)):
            for f in files:
                if not f.endswith((".c", ".cpp", ".h", ".hpp")):
                    continue
                path = os.path.join(d, f)
                rel = os.path.relpath(path, REF)
                for ln, line in enumerate(open(path, errors="replace"), 1):
                    m = re.match(r"(mbe_[A-Za-z0-9_]+)\(", line)
                    if m:
                        defined.add(m.group(1))
                    for m in re.finditer(r"\b(mbe_[A-Za-z0-9_]+)\s*\(", line):
                        # `mbe_soft_bit (*)[23]` is a cast to a pointer-to-array type, not a call
                        if re.match(r"\s*\(\s*\*", line[m.end() - 1:]):
                            types.add(m.group(1))
                            continue
                        calls.setdefault(m.group(1), "%s:%d" % (rel, ln))
    ext = {k: v for k, v in sorted(calls.items()) if k not in defined and k not in types}
    json.dump({"source": "every mbe_* call in the reference's src/ and include/ whose name the reference does not define",
               "symbols": ext}, open(OUT, "w"), indent=1)
    print(len(ext), "symbols ->", OUT)
    for k, v in ext.items():
        print(" ", k, v)


if __name__ == "__main__":
    sys.exit(main())
