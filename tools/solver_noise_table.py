"""profiles/r05_solver_noise.md from gpurun_out/r05_solver_noise.json (tools/solver_noise_probe.py)."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_solver_noise.json"
d = json.load(open(src))
print("# Round 5 - is the parity margin a property of the solver? (`tools/solver_noise_probe.py`)\n")
print("The same ten-step fits (20 k cells, lambda_ = 0.02, the committed oracle fixtures of tests/test_gpu_scale.py) with the\n"
      "truncated minimum-norm solve evaluated seven ways that agree to ~1e-6 on any ONE system.  Entries: deviation from the\n"
      "oracle as a multiple of the reference's own floor (lstsq -> eigh, Gram summation order; tests/_floors.py).\n")
for case, rec in d.items():
    print(f"## {case}\n")
    qs = list(next(iter(rec["variants"].values())).keys())
    print("| solver | " + " | ".join(f"{q} x floor" for q in qs) + " |")
    print("|---|" + "---|" * len(qs))
    for label, row in rec["variants"].items():
        print(f"| {label} | " + " | ".join(f"{row[q]['x']:.2f}" for q in qs) + " |")
    print(f"\nfield: min {rec['V_x_min']:.2f}, mean {rec['V_x_mean']:.2f}, max {rec['V_x_max']:.2f} x floor\n")
print("Reading: the spread between equivalent solvers (M = 3000, float64: 0.88 - 1.19 x) is as large as the move VERDICT r4\n"
      "flagged (1.02 x with the Jacobi solve -> 1.19 x with the deflated one): the trajectories of the ill-conditioned EM\n"
      "iteration amplify a 1e-6 difference of one solve into a different draw from the same noise ball within a few steps.\n"
      "No variant is systematically closer to the oracle across the four cases.  Round 5 nevertheless runs three applications\n"
      "of S2^-1 for the 256-vector block (0.3 ms of a 7 ms solve): it is the variant with the provably smaller subspace error.")
