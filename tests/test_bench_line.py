"""CPU checks of what bench.py prints and of how it starts N > 1 ranks.

The driver reads the LAST stdout line of `python3 bench.py --gpus N --steps K --warmup W` as one JSON object; round 3's line
had grown to 22 KB and the record's `parsed` came back null.  The line is now the compact form of the result
(bench.compact_line) and everything else goes to bench_detail.json / stderr: here a worst-case result (round 3's real
22 KB line with every free-text field blown up, every leg present, NaNs in it) must still give a strict-JSON line under
8 KB that carries every key the contract names.  And `bench.py --gpus 2` with no launcher must start its own ranks as a
child process and hand back rank 0's line (rehearsed on gloo with the stub workload: no filter engine, no GPU)."""
import copy
import json
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def strict_loads(line):
    def no_constants(name):
        raise ValueError(f"non-finite constant {name} in the line")
    return json.loads(line, parse_constant=no_constants)


def r03_result():
    """round 3's real line (the one the driver could not parse), as the full `out` dict of that run"""
    return json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))


def blow_up(out):
    """every string 20 x longer, every list 10 x longer, more legs than any run has, NaN / inf where floats are"""
    def grow(x):
        if isinstance(x, str):
            return x * 20
        if isinstance(x, list):
            return [grow(v) for v in x] * 10
        if isinstance(x, dict):
            return {k: grow(v) for k, v in x.items()}
        return x
    big = grow(copy.deepcopy(out))
    for i in range(12):
        big["workloads"][f"another_leg_with_a_long_name_{i}"] = copy.deepcopy(big["workloads"]["long"])
    big["workloads"]["broken"] = {"error": "RuntimeError('" + "x" * 5000 + "')"}
    big["roofline"]["traffic_rate_GBps"] = float("nan")
    big["roofline"]["frac_of_probe_ceiling"] = float("inf")
    big["cpu_baseline_error"] = "MemoryError " * 500
    return big


def check_line(line, n_gpus=1, unit="Mbp/s"):
    assert "\n" not in line
    d = strict_loads(line)
    for k in CONTRACT_KEYS:
        assert k in d, k
    for k in ROOFLINE_KEYS:
        assert k in d["roofline"], k
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["unit"].startswith(unit)
    return d


def test_round3s_result_gives_a_line_under_the_target():
    out = r03_result()
    line = json.dumps(bench.compact_line(out), allow_nan=False, separators=(",", ":"))
    assert len(json.dumps(out)) > 20_000          # what the driver could not take
    assert len(line) < bench.LINE_TARGET, len(line)
    d = check_line(line)
    assert abs(d["value"] - out["value"]) <= 1e-5 * out["value"]
    assert abs(d["roofline"]["frac"] - out["roofline"]["frac"]) <= 1e-5 * out["roofline"]["frac"]
    assert d["roofline"]["traffic"] == pytest.approx(out["roofline"]["traffic"], rel=1e-5)
    assert d["cpu_baseline"]["cores"] == out["cpu_baseline"]["cores"] and d["cpu_baseline"]["kind"] == "port"
    assert d["cpu_baseline"]["decisions_match_gpu"] is True
    # one scalar per extra leg + whether the oracle agreed
    for leg in ("long", "paired", "union950m"):
        assert d["legs"][leg]["value"] == pytest.approx(out["workloads"][leg]["value"], rel=1e-5)
        assert d["legs"][leg]["decisions_match"] is True
    for kind in ("pageable", "pinned", "packed"):
        assert d["host_path"][kind]["value"] == pytest.approx(out["host_path"][kind]["value"], rel=1e-5)
    for leg in ("search50", "deplete95", "paired"):
        assert d["cli"][leg]["value"] == pytest.approx(out["cli"][leg]["Mbp_per_s_incl_index_load"], rel=1e-5)
    assert d["all_checks_ok"] is True and d["checks"] >= 10


def test_worst_case_result_stays_under_the_cap(tmp_path, capsys):
    big = blow_up(r03_result())
    assert len(json.dumps(big)) > 200_000
    line = bench.emit(big, str(tmp_path / "detail.json"))
    assert len(line) < bench.LINE_CAP, len(line)
    d = check_line(line)
    assert d["roofline"]["frac_of_probe_ceiling"] is None       # inf -> null, never a bare Infinity
    assert d["all_checks_ok"] is False                           # the leg that failed is counted
    assert d["detail"] == "detail.json"
    captured = capsys.readouterr()
    assert captured.out.strip().splitlines()[-1] == line        # the compact line is the last thing on stdout
    assert captured.err.startswith("[bench detail] ")           # the whole result: one line on stderr ...
    whole = json.load(open(tmp_path / "detail.json"))           # ... and the detail file
    assert whole["workloads"].keys() == big["workloads"].keys()


def test_a_failed_baseline_still_gives_the_contract_keys():
    out = r03_result()
    out["cpu_baseline"] = None
    out["cpu_baseline_error"] = "MemoryError()"
    out.pop("workloads"), out.pop("host_path"), out.pop("cli")
    line = json.dumps(bench.compact_line(out), allow_nan=False)
    d = check_line(line)
    assert d["cpu_baseline"] is None and d["cpu_baseline_error"] == "MemoryError()"


def run_bench(args, env_extra, timeout=300):
    env = dict(os.environ, DCN_BENCH_STUB="1", DCN_BENCH_BACKEND="gloo", DCN_BENCH_NO_BIND="1", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_its_own_ranks(tmp_path):
    detail = tmp_path / "d.json"
    p = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--reads", "1000", "--detail", str(detail)], {})
    assert p.returncode == 0, p.stderr[-3000:]
    assert "starting -m torch.distributed.run" in p.stderr
    d = check_line(p.stdout.strip().splitlines()[-1], n_gpus=2)
    assert d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    col = d["collective"]
    assert col["world_size"] == 2 and col["total_bp_matches"] is True and col["total_bp_all_reduced"] == 2 * 4 * 1000 * 150
    assert json.load(open(detail))["collective"]["backend"] == "gloo"


def test_gpus_1_runs_in_this_process_and_a_failing_child_fails_the_parent(tmp_path):
    p = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0", "--reads", "10", "--detail", ""], {})
    assert p.returncode == 0, p.stderr[-3000:]
    assert "torch.distributed.run" not in p.stderr
    check_line(p.stdout.strip().splitlines()[-1], n_gpus=1)
    # the child's exit code is the parent's: an argument the ranks refuse
    p = run_bench(["--gpus", "2", "--no-such-flag"], {})
    assert p.returncode != 0


def test_self_launch_decides_before_any_gpu_library_is_loaded():
    """the launcher parent must not have made a GPU call: the decision runs above the torch / library imports"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("self_launch_if_needed(sys.argv[1:])") < src.index("import torch")
    assert "os.exec" not in src                               # a child process, never a replacement of this one
    assert bench.self_launch_if_needed(["--gpus", "1"]) is None
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert bench.self_launch_if_needed(["--gpus", "2"]) is None   # already under a launcher
    finally:
        del os.environ["WORLD_SIZE"]
