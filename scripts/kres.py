#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin): one line per kernel."""
import re, subprocess, sys
name, d = None, {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name, d = m.group(1), {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
        m2 = re.search(re.escape(key) + r": ([0-9]+)", ln)
        if m2:
            d[key.split(" ")[0]] = int(m2.group(1))
    if "LDS Size" in ln and name:
        try:
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dn = name
        dn = re.sub(r"\(.*", "", dn)
        print(f"{dn[:120]:120s} vgpr {d.get('VGPRs')} agpr {d.get('AGPRs')} scratch {d.get('ScratchSize')} occ {d.get('Occupancy')} lds {d.get('LDS')} sgpr {d.get('SGPRs')}")
