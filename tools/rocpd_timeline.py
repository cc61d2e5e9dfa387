"""Per-launch timeline of a rocprofv3 (rocpd sqlite) kernel trace: name, start offset, duration, gap to the previous kernel.

    python tools/rocpd_timeline.py trace.db [first_kernel_substring] [count]

Prints the launches from the LAST occurrence of `first_kernel_substring` (default: assemble_kernel = the start of one
minimum-norm solve) for `count` kernels, plus the sums of kernel time and of the gaps between them.
"""
import sqlite3
import sys


def main(path, first="assemble_kernel", count=200):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    if not idx:
        print("no kernel matching", first)
        return
    i0 = idx[-1]
    sel = rows[i0:i0 + count]
    t0 = sel[0][1]
    busy = gaps = 0.0
    prev_end = None
    print("| # | kernel | start us | dur us | gap us |")
    print("|---|---|---|---|---|")
    for j, (name, s, e) in enumerate(sel):
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        short = name.split("(")[0].replace("void ", "").replace("mvf::", "")[:44]
        print(f"| {j} | `{short}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} |")
        prev_end = e
    print(f"\n{len(sel)} launches: kernel time {busy:.1f} us, gaps {gaps:.1f} us, span {(sel[-1][2] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3] or ["assemble_kernel"]), *(int(a) for a in sys.argv[3:4]))
