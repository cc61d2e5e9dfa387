#!/usr/bin/env python
"""Turn the parity lines that tests/test_gpu_scale.py prints (pytest -s) into the markdown table of DESIGN.md 2.1 / profiles/rNN_parity_table.md.

    python tools/parity_table.py gpurun_out/r3b/tests.log
"""
import re
import sys

rows = []
for line in open(sys.argv[1], errors="replace"):
    line = line.lstrip(".FEsx")
    m = re.match(r"(C2 lambda [\d.]+|C5 organ|M=\d+ lambda=[\d.]+|C4 generator .*?lambda=[\d.]+|c3_full \(.*?\)|c4_rank \(.*?\)|"
                 r"c4_step \(.*?\)|c4_3step \(.*?\)|PIVOT [\w ,]+?) (float\d+): (?:iterations \d+; )?(.*)", line)
    if not m:
        continue
    case, dtype, rest = m.groups()
    cells = {}
    for q, gpu, floor, ratio in re.findall(r"(\w+) gpu ([\d.e+-]+) / floor ([\d.e+-]+) \(x([\d.]+)", rest):
        cells[q] = (gpu, floor, ratio)
    if cells:
        rows.append((case, dtype, cells))
cols = ["V", "hull", "grid12", "grid", "sigma2", "P999", "P", "E"]
print("Asserted at 1.25 x floor: V, hull, grid12 (grid within 1.2 hull radii), sigma2, P999 (99.9th percentile of |dP|), E. "
      "Reported only: grid (whole bounding box), P (max |dP|) where P999 is present.\n")
print("| case | mode | " + " | ".join(f"{c}: gpu / floor (×)" for c in cols) + " |")
print("|---|---|" + "---|" * len(cols))
for case, dtype, cells in rows:
    out = []
    for c in cols:
        if c in cells:
            g, f, r = cells[c]
            out.append(f"{float(g):.1e} / {float(f):.1e} ({r})")
        else:
            out.append("—")
    print(f"| {case} | {dtype} | " + " | ".join(out) + " |")
