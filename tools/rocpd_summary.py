"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import sqlite3
import sys


def main(path, by_grid=False):
    """by_grid: one row per (kernel, grid size) - tells the launches of the same kernel at different problem sizes apart."""
    con = sqlite3.connect(path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels "
        f"group by name{', grid_x' if by_grid else ''} order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % | vgpr | agpr | sgpr | lds B | grid_x | wg_x |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e6:.4f} | {r[4]/1e6:.4f} | {r[5]/1e6:.4f} | "
              f"{100*r[2]/total:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main(sys.argv[1], by_grid="--by-grid" in sys.argv[2:])
