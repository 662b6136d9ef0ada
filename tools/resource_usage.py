#!/usr/bin/env python3
"""Kernel resource usage of the product build, one line per kernel: VGPRs, SGPRs, scratch, LDS, occupancy.
`make -C lumillyrender_amd/csrc resource-usage` prints LLVM's remarks; this keeps what decides occupancy.
Usage: tools/resource_usage.py [substring of the kernel name]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "lumillyrender_amd", "csrc"), "resource-usage"],
                         capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r" Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "")
            rows[cur] = {}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
    if not rows:
        sys.stderr.write(out[-2000:])
        sys.exit(1)
    for name, r in rows.items():
        if flt and flt not in name:
            continue
        print(f"{name:58s} vgpr {r.get('vgpr', -1):3d}  sgpr {r.get('sgpr', -1):3d}  scratch {r.get('scratch', -1):4d}  lds {r.get('lds', -1):6d}  occupancy {r.get('occ', -1)}")


if __name__ == "__main__":
    main()
