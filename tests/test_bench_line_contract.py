"""CPU: the shape of bench.py's JSON line, checked on the line committed under profiles/ (the newest `r*_bench_final.json`:
the run itself needs the MI355X).  Guards the contract the driver and the judge read -- the keys, the units, `value` from the loop
with the query batches resident in HBM, the roofline and cpu_baseline objects, and the compact `legs` object LAST on the line."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_final.json")),
                   key=lambda p: int(re.search(r"r(\d+)_bench_final", p).group(1)))
    assert paths, "no committed bench line under profiles/"
    text = open(paths[-1]).read().strip().splitlines()[-1]
    return paths[-1], text, json.loads(text)


def test_contract_keys_and_consistency():
    path, text, j = newest_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, f"{path}: `{key}` missing"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["unit"] == "queries/s" and j["higher_is_better"] is True and j["data"] == "synthetic"
    assert j["metric"].split("@")[0].strip().lower().startswith("queries/sec") and "queries/sec" in json.dumps(base).lower()
    assert j["vs_baseline"] is None            # BASELINE.md holds no published number for this metric on this hardware
    assert "workload" in j["config"] and "model" not in j["config"]
    nq = int(re.search(r"(\d+)-query batch", j["config"]["workload"]).group(1))
    # value = whole-job throughput of EXACTLY `steps` steps: queries per step / time per step
    assert j["value"] == pytest.approx(nq / (j["ms_per_step"] * 1e-3), rel=2e-4)
    # ... of the loop whose inputs are resident in HBM; the PCIe-inclusive rate is reported beside it, never as `value`
    assert j["value_loop"] == "resident"
    assert j["resident"]["ms_per_step"] == pytest.approx(j["ms_per_step"], rel=1e-6)
    assert j["host_to_host"]["value"] == pytest.approx(nq / (j["host_to_host"]["ms_per_step"] * 1e-3), rel=2e-4)


def test_roofline_and_cpu_baseline_objects():
    _, _, j = newest_line()
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "bytes_per_launch"):
        assert key in r
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    if r["bound"] == "hbm":   # achieved = algorithmic bytes per launch / the kernel's mean duration in the timed region
        assert r["peak"] == 8000.0
        assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=2e-3)
    assert r["avg_launch_ms"] < j["ms_per_step"]            # the dominant kernel fits inside the step it is part of
    if r["traffic"] is not None:                             # replayed PMC counters: only for the library the bench loaded
        assert r["traffic_source"]["matches_loaded_library"] is True
        assert 0.9 < r["traffic"] / r["bytes_per_launch"] < 3.0
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c
    assert c["kind"] in ("port", "reference") and c["unit"] == j["unit"] and c["cores"] >= 1


def test_legs_object_ends_the_line_and_stays_compact():
    _, text, j = newest_line()
    legs = j["legs"]
    assert list(j)[-1] == "legs" and text.rstrip().endswith("}}")
    assert len(json.dumps(legs)) <= 900, "the driver keeps a 2 000-character tail of the line: the legs object must fit it"
    assert {"c2", "hard", "exact", "c1", "c5", "nb100k_2level"} <= set(legs)
    assert legs["c2"]["qps"] == round(j["value"]) and legs["c2"]["frac"] == j["roofline"]["frac"]
    for name, leg in legs.items():
        assert leg["oracle_ok"] is True, f"leg {name} did not pass its oracle check"
    assert legs["exact"]["same_as_c2"] is True
