"""CPU-side checks (no GPU): liblvae.so loads and exports every symbol include/lvae.h declares, the
host-only parts of the C ABI (workspace queries, argument validation, error text) behave, and the
Python mirror refuses to run without the HIP path (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lvae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lvae_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from lvae import _native as N
    lib = N.lib()
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/lvae.h but not exported by liblvae.so"
    assert sorted(N.exported_symbols()) == decl, "ctypes signature table and header disagree"
    assert lib.lvae_version() >= 100


def test_workspace_queries_are_host_only():
    from lvae import _native as N
    lib = N.lib()
    out = ctypes.c_int64(0)
    assert lib.lvae_log_qz_workspace_bytes(65536, 65536, 32, 0, out) == 0
    base = out.value
    assert base > 2 * 65536 * 64 * 4 and base % 256 == 0
    assert lib.lvae_log_qz_workspace_bytes(65536, 65536, 32, 1, out) == 0
    assert out.value > base                      # prod-marginals partials need more scratch
    assert lib.lvae_log_qz_workspace_bytes(1, 1, 1, 0, out) == 0 and out.value > 0
    assert lib.lvae_log_qz_workspace_bytes(100, 77, 200, 0, out) == 0     # D > 128: direct path only
    assert lib.lvae_mmd_workspace_bytes(4096, 1000, 32, out) == 0 and out.value > 0
    assert lib.lvae_pair_bounds_workspace_bytes(128, 32, out) == 0 and out.value > 0


def test_argument_validation_and_error_text():
    from lvae import _native as N
    lib = N.lib()
    out = ctypes.c_int64(0)
    assert lib.lvae_log_qz_workspace_bytes(0, 10, 32, 0, out) == -1
    assert b"bad shape" in lib.lvae_last_error()
    assert lib.lvae_log_qz_workspace_bytes(10, 10, 32, 0, None) == -1
    assert lib.lvae_pair_bounds_workspace_bytes(1, 32, out) == -1           # MBU needs K >= 2
    # NULL data pointers are rejected before anything touches the device
    rc = lib.lvae_log_qz_f32(None, None, None, 0, 8, 8, 32, 0.0, 0, 0, 0, None, None, None, 0, None)
    assert rc == -1
    rc = lib.lvae_reparam_iw_f32(None, None, None, None, 4, 4, 32, None, None, None, None, None)
    assert rc == -1 and b"non-NULL" in lib.lvae_last_error()
    rc = lib.lvae_kde_score_f32(None, None, 4, 0, 4, 1, 1.0, 0, None, None, 0, None)
    assert rc == -1
    dummy = ctypes.c_void_p(256)
    arr = (ctypes.c_float * 2)(0.1, -1.0)
    rc = lib.lvae_mmd_rbf_f32(dummy, dummy, 8, 8, 32, arr, 2, 1e-5, 1, 0, dummy, dummy, 1 << 30, None)
    assert rc == -1 and b"non-negative" in lib.lvae_last_error()
    rc = lib.lvae_mmd_rbf_f32(dummy, dummy, 8, 8, 5000, arr, 1, 1e-5, 1, 0, dummy, dummy, 1 << 30, None)
    assert rc == -1 and b"D <= 4096" in lib.lvae_last_error()
    ok = (ctypes.c_float * 2)(0.1, 1.0)
    # the MMD entry points take their contraction per call: unknown paths and unsupported combinations are refused
    rc = lib.lvae_mmd_rbf_f32(dummy, dummy, 8, 8, 32, ok, 2, 1e-5, 1, 9, dummy, dummy, 1 << 30, None)
    assert rc == -1 and b"unknown path" in lib.lvae_last_error()
    rc = lib.lvae_mmd_rbf_f32(dummy, dummy, 8, 8, 300, ok, 2, 1e-5, 1, 3, dummy, dummy, 1 << 30, None)
    assert rc == -4 and b"D <= 256" in lib.lvae_last_error()
    rc = lib.lvae_kernel_sum_f32(dummy, dummy, 8, 8, 200, ok, 2, 1e-5, 0, 0, 0, 3, dummy, None, dummy, 1 << 30, None)
    assert rc == -4 and b"split-bf16" in lib.lvae_last_error()
    rc = lib.lvae_mmd_perm_f32(dummy, 8, 0, dummy, 8, 32, dummy, 4, ok, 2, 1e-5, 0, 2, dummy, dummy, 1 << 30, None)
    assert rc == -4 and b"direct-difference" in lib.lvae_last_error()
    assert lib.lvae_dev_reload_env() == 0
    # workspace checks: missing / misaligned / too small
    rc = lib.lvae_log_qz_f32(dummy, dummy, dummy, 0, 8, 8, 32, 0.0, 0, 0, 0, dummy, None, None, 0, None)
    assert rc == -2 and b"NULL" in lib.lvae_last_error()
    rc = lib.lvae_log_qz_f32(dummy, dummy, dummy, 0, 8, 8, 32, 0.0, 0, 0, 0, dummy, None,
                             ctypes.c_void_p(260), 1 << 30, None)
    assert rc == -2 and b"aligned" in lib.lvae_last_error()
    rc = lib.lvae_log_qz_f32(dummy, dummy, dummy, 0, 8, 8, 32, 0.0, 0, 0, 0, dummy, None, dummy, 16, None)
    assert rc == -2 and b"too small" in lib.lvae_last_error()
    rc = lib.lvae_log_qz_f32(dummy, dummy, dummy, 0, 8, 8, 32, 0.0, 0, 0, 7, dummy, None, dummy, 1 << 30, None)
    assert rc == -1 and b"path" in lib.lvae_last_error()
    rc = lib.lvae_log_qz_f32(dummy, dummy, dummy, 0, 8, 8, 200, 0.0, 0, 0, 1, dummy, None, dummy, 1 << 30, None)
    assert rc == -4 and b"D <= 128" in lib.lvae_last_error()


def test_encode_tail_entry_points_validate_on_the_host():
    """lvae_encode_tail_ws_f32 / _bwd_ws_f32: envelope, form selection, workspace size -- all decided before a launch."""
    from lvae import _native as N
    lib = N.lib()
    out = ctypes.c_int64(0)
    # one workgroup up to B = 128 (B * D <= 4096), the multi-workgroup form up to B = 4096 at D <= 64
    assert lib.lvae_encode_tail_supported(32, 1, 32) == 1 and lib.lvae_encode_tail_supported(256, 1, 32) == 0
    assert lib.lvae_encode_tail_ws_supported(256, 1, 32) == 1 and lib.lvae_encode_tail_ws_supported(4096, 1, 64) == 1
    assert lib.lvae_encode_tail_ws_supported(4097, 1, 32) == 0 and lib.lvae_encode_tail_ws_supported(256, 1, 65) == 0
    assert lib.lvae_encode_tail_ws_supported(16, 1, 200) == 1            # wide latents: the one-workgroup form only
    assert lib.lvae_encode_tail_workspace_bytes(512, 1, 32, out) == 0
    need = out.value
    assert need >= 512 * 512 * 4 + 3 * 512 * 32 * 4 and need % 256 == 0  # the [B, B] weights + three [B, D] arrays
    assert lib.lvae_encode_tail_workspace_bytes(16, 1, 200, out) == 0 and out.value == 0
    assert lib.lvae_encode_tail_workspace_bytes(0, 1, 32, out) == -1
    dummy = ctypes.c_void_p(256)
    args = (dummy, dummy, dummy, dummy, 512, 1, 32, 0.0, dummy, dummy, dummy, None, None, dummy)
    assert lib.lvae_encode_tail_ws_f32(*args, 0, dummy, need - 1, None) == -2 and b"workspace" in lib.lvae_last_error()
    assert lib.lvae_encode_tail_ws_f32(*args, 0, None, 0, None) == -2
    assert lib.lvae_encode_tail_ws_f32(*args, 3, dummy, need, None) == -1 and b"mode" in lib.lvae_last_error()
    assert lib.lvae_encode_tail_ws_f32(*args, 1, dummy, need, None) == -4         # B = 512 is outside the one-workgroup form
    wide = (dummy, dummy, dummy, dummy, 16, 1, 200, 0.0, dummy, dummy, dummy, None, None, dummy)
    assert lib.lvae_encode_tail_ws_f32(*wide, 2, dummy, 1 << 30, None) == -4      # ... and D = 200 outside the other
    assert lib.lvae_encode_tail_ws_f32(None, dummy, dummy, dummy, 512, 1, 32, 0.0, dummy, None, None, None, None, None,
                                       0, dummy, need, None) == -1
    bargs = (dummy, dummy, dummy, 512, 1, 32, 0.0, None, dummy, None, dummy, dummy)
    assert lib.lvae_encode_tail_bwd_ws_f32(*bargs, dummy, dummy, 0, dummy, need - 1, None) == -2
    assert lib.lvae_encode_tail_bwd_ws_f32(*bargs, None, None, 0, dummy, need, None) == -1   # no output gradient
    assert lib.lvae_encode_tail_bwd_ws_f32(*bargs, dummy, dummy, 1, dummy, need, None) == -4


def test_no_cpu_fallback():
    """The product path must fail loudly instead of silently computing on the CPU."""
    from lvae import _native as N
    from lvae import estimators as E
    from lvae import mutual_information as MI
    mu, lv, z = torch.zeros(4, 32), torch.zeros(4, 32), torch.zeros(4, 32)
    with pytest.raises(N.LvaeError, match="no CPU fallback"):
        E.approximate_log_q_z(mu, lv, z, prod_marginals=True)
    with pytest.raises(N.LvaeError):
        E.sample_log_likelihood(z, mu, lv)
    with pytest.raises(N.LvaeError):
        E.kl_divergence(mu, lv)
    with pytest.raises(N.LvaeError):
        MI.calc_upper_and_lower_bound_representational_mi(z, torch.zeros(4, 64))
    zg = torch.zeros(4, 32, requires_grad=True)
    with pytest.raises(N.LvaeError, match="no CPU fallback"):       # differentiable path: HIP only as well
        E.approximate_log_q_z(zg, lv, z, prod_marginals=True)
    with pytest.raises(N.LvaeError, match="no CPU fallback"):
        E.sample_log_likelihood(zg, mu, lv)
    with pytest.raises(N.LvaeError, match="forward-only"):          # estimators without a backward say so
        MI.calc_upper_and_lower_bound_representational_mi(zg, torch.zeros(4, 64))


def test_missing_library_is_an_error(monkeypatch):
    from lvae import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/liblvae.so")
    with pytest.raises(N.LvaeError, match="missing"):
        N.lib()


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under the package may import it."""
    pkg = os.path.join(ROOT, "language-transformer-vae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle-backed", ""), f"{f} mentions the oracle"


def test_shard_bounds():
    from lvae.distributed import shard_bounds
    for n, ws in ((10, 3), (7, 8), (65536, 8), (1, 1)):
        spans = [shard_bounds(n, ws, r) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_grid_quantisation_of_multi_gpu_shapes():
    """Host logic: the split of the posterior axis must not leave a nearly empty last round of workgroups
    (e.g. 543 equal workgroups on 512 resident slots) for the per-rank shapes of the weak-scaling bench."""
    import ctypes
    import math
    from lvae import _native as N
    for world in (1, 2, 4, 8):
        q = 256 * world
        n = max(q, int(round(65536 * math.sqrt(world) / q)) * q)
        out = (ctypes.c_int64 * 8)()
        N.check(N.lib().lvae_plan_info(n // world, n, 32, 0, out))
        dp, ncb, cpad, nsplit, bps, _, nblk, _ = list(out)
        assert dp == 32 and nsplit * bps >= nblk
        groups = cpad // (128 * ncb)
        rounds = -(-groups * nsplit // 512)
        eff = (groups * nblk / 512) / (rounds * bps)
        assert eff > 0.95, (world, groups, nsplit, eff)


def test_committed_bench_line_has_the_contract_keys():
    """profiles/r0N/bench_n1.json are the lines bench.py printed on the MI355X: the driver's contract keys, the
    roofline object and the CPU baseline must all be there, and the numbers must be self-consistent."""
    for tag in ("r01", "r02"):
        _check_bench_line(os.path.join(ROOT, "profiles", tag, "bench_n1.json"))
    import json
    r02 = json.loads(open(os.path.join(ROOT, "profiles", "r02", "bench_n1.json")).read())
    side = r02["also_measured"]["1m"]            # the north-star size on one GPU rides along
    assert side["N"] == 1048576 and side["scaling"] == "strong"
    assert abs(side["value"] - side["N"] ** 2 / (side["ms_per_step"] * 1e-3)) < 1e-6 * side["value"]


def _check_bench_line(path):
    import json
    line = json.loads(open(path).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"]
    assert "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == line["unit"]
    n = line["config"]["N"]
    assert abs(line["value"] - n * n / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_mmd_workspace_queries_cover_the_layout_of_every_mode():
    """The split search of the pair plan is monotone neither in the shape nor in `symmetric`, so a
    *_workspace_bytes query must cover the extent of the layout each mode of its entry point really uses
    (lvae_layout_info).  Includes the shapes on which the round-1 sizing fell short."""
    from lvae import _native as N
    lib = N.lib()
    rng = np.random.default_rng(7)
    shapes = [(22307, 92021, 8), (43846, 45790, 33), (18782, 185975, 128), (44842, 46005, 128)]
    for _ in range(4000):
        nb = int(rng.integers(2, 300000))
        na = int(rng.integers(1, nb + 1))
        shapes.append((na, nb, int(rng.integers(1, 257))))
    info = (ctypes.c_int64 * 3)()
    out = ctypes.c_int64(0)
    for na, nb, D in shapes:
        N.check(lib.lvae_kernel_sum_workspace_bytes(na, nb, D, out))
        for sym in (0, 1):
            N.check(lib.lvae_layout_info(0, na, nb, D, sym, info))
            assert info[0] <= out.value, ("kernel_sum", na, nb, D, sym)
        N.check(lib.lvae_mmd_perm_workspace_bytes(na, nb, D, out))
        for sym in (0, 1):
            N.check(lib.lvae_layout_info(1, na, nb, D, sym, info))
            assert info[0] <= out.value, ("mmd_perm", na, nb, D, sym)
        if na >= 2:
            N.check(lib.lvae_mmd_workspace_bytes(na, nb, D, out))
            for which in (0, 1, 2):
                N.check(lib.lvae_layout_info(2, na, nb, D, which, info))
                assert info[0] <= out.value, ("mmd_rbf", na, nb, D, which)


def test_pair_bounds_workspace_covers_both_of_its_log_qz_calls():
    """lvae_pair_bounds_f32 runs log q(z) twice -- all pairs, then with the diagonal excluded -- and the two calls plan
    differently at D in (32, 64]; its workspace query must cover the larger plan (K = 20,000 at D = 64 was refused by
    the first call when the region was sized for the excluded-diagonal plan alone)."""
    from lvae import _native as N
    lib = N.lib()
    rng = np.random.default_rng(11)
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    shapes = [(20000, 64), (20000, 32), (20000, 128), (2, 1), (333, 768)]
    shapes += [(int(rng.integers(2, 60000)), int(rng.integers(1, 200))) for _ in range(3000)]
    for K, D in shapes:
        N.check(lib.lvae_pair_bounds_workspace_bytes(K, D, a))
        N.check(lib.lvae_log_qz_workspace_bytes(K, K, D, 0, b))
        assert a.value >= b.value + 3 * 4 * K, (K, D, a.value, b.value)


def test_backward_workspace_query_covers_both_gradient_modes():
    """lvae_log_qz_bwd_f32 picks its path from the gradients it is given; a workspace sized with prod_marginals = 1
    must also serve the call that passes g_log_qz alone (the matrix-core sweeps' region)."""
    from lvae import _native as N
    lib = N.lib()
    for M, Nn, D in ((32, 32, 32), (300, 7, 5), (4096, 4096, 64), (20000, 20000, 96), (16384, 16384, 128), (100, 100, 200)):
        a, b = N._i64(0), N._i64(0)
        N.check(lib.lvae_log_qz_bwd_workspace_bytes(M, Nn, D, 0, a))
        N.check(lib.lvae_log_qz_bwd_workspace_bytes(M, Nn, D, 1, b))
        assert b.value >= a.value > 0, (M, Nn, D)
