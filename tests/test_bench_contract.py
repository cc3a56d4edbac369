"""The bench.py contract (one JSON line; keys the driver and the judge read), checked on the line
committed under profiles/ by the last GPU run, plus the host-side logic of the multi-rank rehearsal."""

import importlib.util
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _committed_line():
    for name in ("r03_bench_n136.json", "r02_bench_n136.json", "r01_bench_n136.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            return path
    raise FileNotFoundError("no committed bench line under profiles/")


def test_committed_bench_line_has_the_contract_keys():
    with open(_committed_line()) as fh:
        lines = [ln for ln in fh.read().splitlines() if ln.strip()]
    assert len(lines) == 1                                   # ONE line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "iters/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.5 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0.5 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 2 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "iters/s" and c["value"] > 0 and c["sample"]
    assert d["parity"]["history_max_rel_diff"] < 1e-8
    if "window_values" in d:                                 # round 3: value = median of consecutive windows of `steps` steps
        import statistics
        assert len(d["window_values"]) >= 5 and abs(statistics.median(d["window_values"]) - d["value"]) < 1e-6 * d["value"]
        assert max(d["window_values"]) - min(d["window_values"]) < 0.05 * d["value"]
        for key in ("roofline_mypre_a_gs", "secondary_configs", "roofline_hdg_like"):
            assert key in d and "error" not in (d[key] or {}), key
        m = d["roofline_mypre_a_gs"]
        assert m["valid"] and 0.3 < m["sweep_call"]["frac"] < 1.0 and 0.3 < m["auxiliary_space_term"]["frac"] < 1.0
        sec = d["secondary_configs"]
        assert sec["cfg2"]["minres"]["us_per_iteration"] < 60 and sec["cfg3"]["bpcg_v2"]["us_per_iteration"] < 150
        assert "spmv_A_plain_in_loop_cache_state" in d["hbm_GBs"]


def test_rehearsal_reads_the_markers_of_the_child(tmp_path, monkeypatch):
    """`rehearse_native_path`: the fastest data path whose marker the child printed wins; a child
    that hangs after confirming a path keeps that path; no marker -> torch.distributed."""
    bench = _bench()
    script = tmp_path / "child.py"
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setenv("NSS_PROBE_FORCE", "1")
    monkeypatch.setenv("MASTER_PORT", "29500")
    args = types.SimpleNamespace(gpus=2, pre="bjac3")

    def run(body, timeout="20"):
        script.write_text("import sys, time\n" + body)
        monkeypatch.setenv("NSS_PROBE_TIMEOUT", timeout)
        return bench.TIERS[bench.rehearse_native_path(args, 0)]

    assert run("print('%srccl-python', file=sys.stderr)\nprint('%snative', file=sys.stderr)\n"
               % (bench.TIER_MARK, bench.TIER_MARK)) == "native"
    assert run("print('%srccl-python', file=sys.stderr, flush=True)\ntime.sleep(30)\n" % bench.TIER_MARK, "2") == "rccl-python"
    assert run("print('nothing useful', file=sys.stderr)\nsys.exit(3)\n") == "torch"
    monkeypatch.delenv("NSS_PROBE_FORCE")
    monkeypatch.setenv("NSS_DIST_BACKEND", "gloo")
    assert run("raise SystemExit('must not be started')\n") == "torch"      # not an RCCL run: no child at all
