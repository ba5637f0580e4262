#!/usr/bin/env python3
"""Build check: the generic scan kernels decode their step program from tables held in VGPR lanes (v_readlane).  A table
the register allocator spills is reloaded only in the lanes active at the reload, so a spill can silently corrupt the
decode: fail the build when a Scan*Kernel spills VGPRs.  Input: the compiler's -Rpass-analysis=kernel-resource-usage
remarks of one source file."""
import re
import sys

bad = []
name = None
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
    m = re.search(r"VGPRs Spill: (\d+)", line)
    if m and name and re.search(r"Scan(Keyed|NoKey|Filter|Project)Kernel", name) and int(m.group(1)) > 0:
        bad.append((name, int(m.group(1))))
for name, n in bad:
    print(f"{sys.argv[1]}: {name} spills {n} VGPRs", file=sys.stderr)
sys.exit(1 if bad else 0)
