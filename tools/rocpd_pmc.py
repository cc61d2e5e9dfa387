"""Per-kernel PMC sums from a rocprofv3 (rocpd sqlite) counter-collection run:
    python tools/rocpd_pmc.py <results.db> [kernel-name-substring]
prints | kernel | counter | launches | avg value per launch | (raw counter units; FETCH_SIZE / WRITE_SIZE are in KB)."""
import sqlite3
import sys


def rows(path, like=""):
    con = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
         "where kernel_name like ? group by kernel_name, counter_name order by sum(value) desc")
    return con.execute(q, (f"%{like}%",)).fetchall()


if __name__ == "__main__":
    print("| kernel | counter | launches | avg raw per launch | raw x 1024 (bytes) |")
    print("|---|---|---|---|---|")
    for name, ctr, n, tot in rows(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")[:40]:
        nm = name if len(name) < 70 else name[:67] + "..."
        print(f"| `{nm}` | {ctr} | {n} | {tot / n:.4g} | {tot / n * 1024:.4g} |")
