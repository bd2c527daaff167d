"""CPU: the line bench.py prints last on stdout is what the driver parses out of an 8 KB tail -- it must stay short and strict JSON.
(Round 5's 22.9 KB line came back as `parsed: null`: the whole round's headline counted as unmeasured.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_constants(name):
    raise AssertionError(f"non-JSON constant {name} in the bench line")


def test_compact_line_from_a_full_result_is_short_strict_json_with_the_contract_fields():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))      # a complete result of the default run (22.9 KB)
    full["round_sweep"] = {"unit": "Msamples/s by fft1 blocks per round", "256": 20000.0, "1024": 30000.0, "4096": 41000.0, "8192": 45000.0}
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < bench.LINE_LIMIT <= 6000
    out = json.loads(line, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["value"] == full["value"] and out["config"]["workload"].startswith("BASELINE configs[2]") and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for k in ("traffic", "alg_bytes_per_launch", "avg_launch_us", "kernel"):
        assert k in r, k
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] in ("reference", "port")
    assert out["full_rings"]["value"] == full["full_rings"]["value"] and out["secondary"]["value"] == full["secondary"]["value"]
    assert out["round_sweep"]["8192"] == 45000.0
    g = out["glue"]
    assert any(isinstance(v, dict) and set(v) == {"0", "1", "2", "3", "4"} for v in g.values())


def test_compact_line_drops_optional_objects_rather_than_grow_past_the_limit():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    full["glue"] = {"error": "x" * 20000}
    line = bench.compact_line(full)
    assert len(line.encode()) < bench.LINE_LIMIT
    out = json.loads(line, parse_constant=_no_constants)
    assert "glue" not in out and out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0
