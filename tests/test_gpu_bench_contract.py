"""The driver's contract for bench.py (one JSON line on rank 0, the keys the judge reads): a short run of
the headline workload and of the configs[4] slice, parsed and checked."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, extra_env=None):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd="/tmp",
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(j, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == steps and j["warmup"] == warmup
    assert j["unit"] == "env-steps/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert isinstance(j["config"]["workload"], str) and "model" not in j["config"]
    E = j["config"]["envs_per_gpu"]
    assert abs(j["value"] - E * steps / (j["ms_per_step"] * 1e-3 * steps)) / j["value"] < 1e-9
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    # the kernel's launch time (HIP events) cannot exceed the wall time per step
    assert r["avg_launch_us"] <= j["ms_per_step"] * 1e3 * 1.001
    assert abs(r["achieved"] - r["bytes_per_env_step"] * E / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-9
    assert "hbm_frac" in r and "wall_frac" in r and r["wall_frac"] <= r["frac"] * 1.001


def test_headline_line_with_cpu_baseline():
    j = _bench("--gpus", "1", "--steps", "20", "--warmup", "5")
    _check_common(j, 20, 5)
    assert j["config"]["envs_per_gpu"] == 65536 and j["config"]["tickers"] == 30
    assert "DOW30" in j["metric"] and j["roofline"]["bytes_per_env_step"] == 1593
    assert j["roofline"]["traffic"] is not None and j["roofline"]["hbm_frac"] > j["roofline"]["frac"]
    assert j["prewarm_launches"] == 2048
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "env-steps/s" and cb["value"] > 1e5
    assert isinstance(cb["sample"], str)
    ps = cb["parity_sample"]
    assert ps["reward_max_abs_delta"] == 0.0 and ps["holdings_match_rate"] == 1.0 and ps["observations_identical"]
    assert j["value"] > 1e7                                  # BASELINE.json's target: >= 10 M env-steps/s


def test_configs4_slice_line():
    j = _bench("--env", "crypto", "--envs-per-gpu", "32768", "--rollout", "16", "--steps", "64", "--warmup", "16",
               "--no-cpu-baseline")
    _check_common(j, 64, 16)
    assert j["config"]["rollout_n_steps"] == 16 and "hipGraph" in j["config"]["launch"]
    assert "cpu_baseline" not in j


def test_two_rank_rehearsal_of_the_scaling_command():
    """`python bench.py --gpus 2` exactly as the driver would issue it for a scaling run, on this one-GPU box:
    bench.py starts its own two ranks (child torch.distributed.run), each builds its shard with the real
    kernels and the timed region gathers episode returns -- with every rank on cuda:0 and gloo instead of RCCL
    (FINENV_BENCH_REHEARSAL=1).  Checks the plumbing and the accounting, not the speed."""
    j = _bench("--gpus", "2", "--envs-per-gpu", "8192", "--steps", "40", "--warmup", "8", "--prewarm", "16",
               "--no-cpu-baseline", extra_env={"FINENV_BENCH_REHEARSAL": "1"})
    assert j["n_gpus"] == 2 and j["steps"] == 40 and j["scaling"] == "weak"
    assert j["config"]["envs_per_gpu"] == 8192 and j["config"]["global_envs"] == 16384
    assert j["config"]["parallelism"] == "env-shard x2"
    assert j["rccl"]["world"] == 2 and j["rccl"]["gathers"] >= 1 and j["rccl"]["gathered"] == 16384
    assert abs(j["value"] - 16384 * 40 / (j["ms_per_step"] * 1e-3 * 40)) / j["value"] < 1e-9
    assert "REHEARSAL" in j["data"] and "cpu_baseline" not in j
