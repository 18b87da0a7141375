"""GPU: bench.py keeps its contract -- `python bench.py --gpus 1 --steps K --warmup W` (the flags the round-end driver passes)
prints ONE JSON record as the LAST line of stdout with the agreed keys, `value` = whole-job lookups per second over exactly
K timed steps, `roofline` measured live with HIP events, `cpu_baseline` from the oracle port on the host cores; and the N > 1
code path (one rank, --force-sharded) prints the same record shape."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    return json.loads(lines[-1]), lines


def test_bench_line_default_form():
    j, lines = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"])
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1, "one JSON record"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["higher_is_better"] is True
    assert j["unit"] == "lookups/s" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic" and j["scaling"] == "weak"
    assert "workload" in j["config"] and "model" not in j["config"]
    B, T = j["config"]["global_batch"], 26
    assert abs(j["value"] - T * B / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]     # whole-job lookups over the timed region
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_ms"] / 1e6) <= 1e-6 * r["achieved"]
    assert r["bytes_per_launch"] == B * 5852                                           # SURVEY 8(d) x the samples of one launch
    assert r["traffic"] is None or 0.5 * r["bytes_per_launch"] < r["traffic"] < 2.0 * r["bytes_per_launch"]
    assert r["avg_launch_ms"] <= j["ms_per_step"] * 1.001                               # events inside the wall-clock region
    m = r["mfma"]
    assert m["insts_per_launch"] == (B // 16) * 16 * 27 and 0.1 < m["frac_of_157.3"] < 1.0
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "lookups/s" and c["sample"]
    assert j["value"] > 50 * c["value"]
    # side lines the judge reads
    for k in ("two_call_path", "multi_batch", "small_batch", "large_batch", "reduced_precision_tables", "reference_benchmark_shape", "h2d_inclusive", "cache_tier"):
        assert k in j, k
    ct = j["cache_tier"]
    assert ct["roofline"]["frac"] > 0.2 and abs(ct["hit_rate"] - ct["oracle_hit_rate"]) < 0.01
    assert ct["mixed_precision_tiers"]["roofline"]["bytes_per_launch"] == B * 3226


def test_bench_line_sharded_path_one_rank():
    for mode in ("inline", "p2p"):
        j, _ = _run(["--gpus", "1", "--force-sharded", "--exchange-mode", mode, "--steps", "50", "--warmup", "5"], timeout=600)
        assert j["n_gpus"] == 1 and j["steps"] == 50 and j["scaling"] == "weak" and j["value"] > 0
        assert j["config"]["exchange_mode"] == mode and j["config"]["placement"] == "rows+replicate"
        assert j["roofline"] is not None and j["roofline"]["bound"] == "hbm"
    # round 6: the default exchange -- the extension's own ncclAllToAllv on the step's stream -- forced on the one rank a box has;
    # it is verified against all_to_all_single (receive buffers bit for bit) before it is timed
    j, _ = _run(["--gpus", "1", "--force-sharded", "--force-exchange", "--steps", "50", "--warmup", "5"], timeout=600)
    assert j["config"]["exchange_requested"] == "direct" and j["config"]["exchange_mode"] == "direct", j["config"]
    assert j["config"]["direct_a2a"] == "ncclAllToAllv"
    assert j["per_rank"]["tables_owned"] == [5] and len(j["per_rank"]["step_ms"]) == 1
