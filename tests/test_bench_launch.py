"""bench.py launched plain (`python bench.py --gpus 2`): it must start its own ranks, exchange over the
process group, and print exactly one JSON line with the contract keys.  Runs on the CPU over gloo with the
per-rank compute injected from tests/bench_cpu_kernels.py (the HIP kernels need a GPU)."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags, timeout=240):
    env = dict(os.environ, LVAE_BENCH_KERNELS="tests.bench_cpu_kernels", OMP_NUM_THREADS="2", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("gpus", [1, 2])
def test_plain_invocation_launches_its_ranks_and_prints_one_line(gpus):
    n = 192
    p = _run({}, "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--n", str(n), "--d", "8",
             "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]   # gloo's own chatter
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == gpus and line["steps"] == 2 and line["config"]["N"] == n
    assert line["config"]["rows_per_gpu"] == n // gpus and "workload" in line["config"]
    assert abs(line["value"] - n * n / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    # the statistics of the last step are the job's: compare with the oracle on the same synthetic rows
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle_np as onp
    parts = [bench.make_inputs(torch, n, 8, r, gpus, torch.device("cpu")) for r in range(gpus)]
    mu, lv, z = (torch.cat([p_[i] for p_ in parts]).numpy() for i in range(3))
    lq, _ = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
    lp = onp.sample_log_likelihood(z, reduce_batch_dim=False)
    want = float(np.mean(lq) - np.mean(lp))
    assert abs(line["results"]["marginal_kl"] - want) <= 1e-6 * abs(want)


def test_a_failing_rank_fails_the_run():
    p = _run({"LVAE_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "64", "--d", "8",
             "--no-cpu-baseline", timeout=120)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_workload_resolution():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "1"])
    assert bench.resolve_workload(a, 1)[:3] == ("c2", 65536, "none")      # one GPU: neither weak nor strong
    a = bench.parse_args(["--gpus", "8"])
    name, n, scaling, desc = bench.resolve_workload(a, 8)
    assert (name, n, scaling) == ("1m", 1048576, "strong") and "north star" in desc
    a = bench.parse_args(["--gpus", "8", "--d", "64"])
    assert "configs[2]" in bench.resolve_workload(a, 8)[3]
    a = bench.parse_args(["--gpus", "4", "--workload", "weak"])
    name, n, scaling, _ = bench.resolve_workload(a, 4)
    assert scaling == "weak" and abs(n * n / 4 - 65536 ** 2) < 0.01 * 65536 ** 2 and n % (256 * 4) == 0
    assert math.isclose(bench.weak_problem_size(1), 65536)
    a = bench.parse_args(["--gpus", "8", "--workload", "mmd"])
    name, n, scaling, desc = bench.resolve_workload(a, 8)
    assert (name, n, scaling) == ("mmd", 262144, "strong") and "configs[3]" in desc
    a = bench.parse_args(["--gpus", "4", "--workload", "iw"])
    name, n, scaling, desc = bench.resolve_workload(a, 4)
    assert (name, n, scaling) == ("iw", 16384, "strong") and "configs[4]" in desc


def _one_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2])
def test_mmd_workload_two_ranks(gpus):
    """BASELINE configs[3] as a bench workload (`--workload mmd`): both samples sharded, mirrored column chunks per
    rank, all-reduce of the label sums -- rehearsed over gloo at a test size against the float64 oracle."""
    n, D, P = 96, 8, 5
    line = _one_line(_run({}, "--gpus", str(gpus), "--workload", "mmd", "--n", str(n), "--d", str(D), "--perms", str(P),
                          "--steps", "2", "--warmup", "1"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == gpus and line["unit"] == "pooled-entries/s" and line["config"]["n1"] == n
    assert line["scaling"] == ("strong" if gpus > 1 else "none")
    assert abs(line["value"] - (2 * n) ** 2 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle_np as onp
    xs, ys = [], []
    for r in range(gpus):
        g = torch.Generator().manual_seed(4321 + r)
        xs.append(torch.randn(n // gpus, D, generator=g))
        ys.append(0.98 * torch.randn(n // gpus, D, generator=g) + 0.01)
    x, y = torch.cat(xs).numpy(), torch.cat(ys).numpy()
    want = onp.mmd_statistic_tts(x, y, bench.ALPHAS)
    assert abs(line["results"]["statistic"] - want) <= 1e-9 * max(1.0, abs(want))
    assert 0.0 <= line["results"]["pval"] <= 1.0 and line["results"]["ms_label_generation"] >= 0.0
    if gpus > 1:
        ph = line["exchange_phases_ms"]
        assert len(ph["per_rank"]) == gpus and set(ph["max_over_ranks"]) == {"all_gather", "kernels", "all_reduce"}
        assert ph["bytes"]["all_reduce"] == (2 * (P + 1) + 1) * 8


@pytest.mark.parametrize("gpus", [1, 2])
def test_iw_workload_two_ranks(gpus):
    """BASELINE configs[4] as a bench workload (`--workload iw`): sentences sharded, results all-gathered."""
    B, K, D = 24, 7, 8
    line = _one_line(_run({}, "--gpus", str(gpus), "--workload", "iw", "--n", str(B), "--k", str(K), "--d", str(D),
                          "--steps", "2", "--warmup", "1"))
    assert line["n_gpus"] == gpus and line["unit"] == "(sentence,sample)/s" and line["roofline"]["bound"] == "hbm"
    assert abs(line["value"] - B * K / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["results"]["gathered_rows"] == B            # every rank ends up with all sentences' results
    if gpus > 1:
        assert set(line["exchange_phases_ms"]["max_over_ranks"]) == {"kernel", "all_gather"}


def test_default_multi_gpu_line_carries_phases_and_the_other_sharded_configs():
    """`bench.py --gpus 2` as the driver runs it (no workload flag; side runs shrunk by --n): the headline is the row-sharded
    log q(z) job and the line says, per rank, how long the compute stream waited for the all-gather."""
    n = 128
    line = _one_line(_run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", str(n), "--d", "8",
                          "--no-cpu-baseline"))
    ph = line["exchange_phases_ms"]
    assert set(ph["max_over_ranks"]) == {"own_block", "wait_all_gather", "rest_block", "merge_and_sums", "all_reduce"}
    assert len(ph["per_rank"]) == 2 and ph["all_gather_alone_ms"] >= 0.0
    assert ph["bytes"]["all_gather_received_per_rank"] == 2 * n * 8 * 4 and ph["bytes"]["all_reduce"] == 40
