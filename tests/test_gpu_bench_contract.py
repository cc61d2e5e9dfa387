"""bench.py's output contract on a small workload (subprocess, one GPU): ONE JSON line on stdout with the driver's keys, the
`roofline` and `cpu_baseline` objects of this tier, this repository's extras (`parity`, `f64`, `env`) and the figures `config`
carries for the driver's parsed record (`f64_value`, `eval_frac`, `c5_32_organs_wall_s`, `rccl_ranks`, ...)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cells", "150000",
           "--ctrl", "1100", "--cpu-cells", "6000"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("MVF_")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["cells"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < r["frac"] < 1.0 and "traffic" in r and r["avg_kernel_ms"] * 0.999 <= d["ms_per_step"]
    # HBM traffic of the dominant kernel measured in the run itself (two rocprofv3 --pmc child passes), not read from a file
    assert r["traffic_from_committed_profile"] is False and r["traffic"] > 0 and "measured in this run" in r["traffic_source"]
    alg = 2.0 * 4 * d["config"]["cells"] * d["config"]["ctrl_points"]
    assert 0.3 * alg < r["traffic"] < 60.0 * alg and r["traffic_detail"]["launches_per_iteration"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "cells/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    par = d["parity"]
    assert par["em_steps"] == 10 and set(par["f64"]) >= {"V_rel_err", "sigma2_rel_err", "P_max_abs_err", "E_rel_err"}
    assert par["f64"]["V_rel_err"] < max(2.0 * par["floor"]["V"], 1e-5) and par["f32"]["V_rel_err"] < max(2.0 * par["floor"]["V"], 1e-3)
    assert d["f64"]["dtype"] == "f64" and d["f64"]["value"] > 0
    assert d["env"] == {} and d["developer_options"] == {}
    assert d["con_k"]["bound"].startswith("hbm") and 0 < d["con_k"]["frac"] < 1
    ev = d["eval"]
    assert ev["grid_points"] == 64**3 and ev["float32"]["kernel_ms_all_quantities"] < 5.0
    assert ev["float64"]["jacobian_plus_curl_api_wall_ms"] < 100.0
    sc = d["small_configs"]
    assert sc["c2_50k_x_500"]["ms_per_em_step"] < 4.0 and sc["c5_organ_250k_x_500"]["ms_per_em_step"] < 6.0
    assert sc["c5_four_organs"]["identical_to_sequential"] is True and sc["c5_four_organs"]["four_streams_wall_s"] > 0
    wf = d["whole_fit"]
    for cfg in ("c2", "c4"):
        split = wf[cfg]["split_of_a_second_call_with_phase_syncs"]
        assert set(split) >= {"preprocess_s", "upload_and_u_cache_s", "em_s", "download_s", "total_s"} and wf[cfg]["wall_s"] > 0
    assert "pivot_subset" not in d
    cfg = d["config"]   # what the driver's parsed record shows of the other configurations
    assert cfg["f64_value"] == d["f64"]["value"] and 0 < cfg["f64_frac"] < 1 and cfg["solve_avg_ms"] > 0
    assert 0 < cfg["eval_frac"] < 1 and 0 < cfg["eval_frac_f64_cells"] < 1 and cfg["eval_api_wall_ms"] > 0
    assert cfg["c2_ms_per_em_step"] == sc["c2_50k_x_500"]["ms_per_em_step"] and cfg["c5_organ_ms_per_em_step"] > 0
    assert cfg["c5_32_organs_wall_s"] == sc["c5_32_organs"]["wall_s"] > 0 and sc["c5_32_organs"]["organs"] == 32
    assert "c3_ms_per_step" not in cfg   # (config 3 runs only when the workload is the full-size one)
    assert cfg["rccl_ranks"] == 1        # ncclCommCount of the one-rank communicator of the rccl_world1 leg
    # the step's collectives executed on a one-rank RCCL communicator through both back ends
    rc = d["rccl_world1"]
    for coll in ("torch", "mvf"):
        assert "failed" not in rc[coll], rc[coll]
        cm = rc[coll]["comm"]
        assert cm["collectives_per_step"] == 4.0 and cm["ranks"] == 1 and cm["forced_on_one_rank"] is True
        assert cm["allreduce_bytes"] == 1100 * 1101 // 2 * 8 and rc[coll]["ms_per_step"] > 0
        pr = rc[coll]["per_rank"][0]
        assert pr["allreduce_ms"] > 0 and pr["rhs_stats_allreduce_ms"] > 0 and pr["scalar_allreduce_ms"] > 0
    assert "nccl" in rc["torch"]["comm"]["backend"] and "mvf_allreduce_stats" in rc["mvf"]["comm"]["backend"]
