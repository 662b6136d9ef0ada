#!/usr/bin/env python3
"""Where a kernel touches scratch: python tools/scratch_in_loops.py <file.s> <kernel substring>
Lists every scratch load / store of the kernels whose mangled name contains the substring, with its line offset inside the
function and the innermost backward branch that encloses it (a spill inside a loop body is what costs)."""
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+:\s", l)]
    for si, s in enumerate(starts):
        name = lines[s].split(":")[0]
        if sub not in name:
            continue
        e = starts[si + 1] if si + 1 < len(starts) else len(lines)
        body = lines[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
        back = []       # (target line, branch line) of backward branches = loops
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", l)
            if m:
                t = labels.get(m.group(1) or m.group(2))
                if t is not None and t < i:
                    back.append((t, i))
        sc = [(i, l.strip()) for i, l in enumerate(body) if "scratch_" in l]
        print(name[:70], "lines", len(body), "loops", len(back), "scratch", len(sc))
        for i, l in sc:
            enc = [(t, b) for t, b in back if t <= i <= b]
            inner = min(enc, key=lambda x: x[1] - x[0]) if enc else None
            print(f"   {i:6d} {'loop[%d..%d] len %d' % (inner[0], inner[1], inner[1] - inner[0]) if inner else 'outside loops':28s} {l[:90]}")


if __name__ == "__main__":
    main()
