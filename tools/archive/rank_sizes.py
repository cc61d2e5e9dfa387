"""One rank's share of the 2- / 4- / 8-GPU strong-scaling runs of the headline workload, RUN on one GPU (same generator,
cells per rank = 8 M / N, no collectives): python tools/rank_sizes.py [--out profiles/rNN_rank_sizes.json]
Per size: the default mode's step (Gram + solve + rest) and the pivot-subset sub-run's, from bench.py's own JSON line."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
res = {}
for cells in (4_000_000, 2_000_000, 1_000_000):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cells", str(cells), "--no-conk", "--no-f64", "--cpu-cells", "0"],
                       capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        print(p.stderr[-2000:], file=sys.stderr)
        raise SystemExit(f"bench.py failed at {cells} cells")
    d = json.loads(p.stdout.strip().splitlines()[-1])
    for tag, r in (("full", d), ("pivot", d.get("pivot_subset"))):
        if r is None:
            continue
        res[f"{cells}_{tag}"] = {"cells": cells, "gram_mode": r.get("config", {}).get("gram_mode", tag) if tag == "full" else r.get("gram_mode", tag),
                                 "ms_per_step": r["ms_per_step"], "gram_ms": r["roofline"].get("avg_kernel_ms"),
                                 "gram_TF": r["roofline"]["achieved"], "solve_ms": r["solve"]["avg_ms"],
                                 "ctrl_used": r.get("ctrl_used", d["config"].get("ctrl_points"))}
    print(cells, json.dumps({k: v for k, v in res.items() if k.startswith(str(cells))}), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
