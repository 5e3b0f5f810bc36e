#!/usr/bin/env python3
"""Benchmark of the latent-analysis hot path: the aggregate-posterior log q(z) matrix with its streaming
logsumexp, reduced to marginal KL / MI.

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine)

Workloads (`--workload`, default by --gpus):
  c2    BASELINE configs[1]: N = M = 65,536, D = 32 on one MI355X -- the configuration the metric is quoted on;
        the default at --gpus 1.
  1m    the north star's multi-GPU shape: N = M = 1,048,576 posteriors / samples in total, rows sharded over the
        ranks (STRONG scaling: the total is fixed); D = 32 for the metric, `--d 64` is BASELINE configs[2].
        The default at --gpus > 1.
  weak  N = M = 65,536 * sqrt(gpus): every GPU evaluates 65,536^2 pairs whatever the rank count.
  mmd   BASELINE configs[3]: Gaussian-kernel MMD permutation test, n1 = n2 = 262,144, D = 32, 100 permutations; both
        samples sharded by rows over the ranks (fixed total: STRONG scaling) -- all-gather of the samples, every rank
        evaluates two mirrored column chunks of the upper triangle, all-reduce of 2P + 1 float64 sums.
  iw    BASELINE configs[4]: IW log-likelihood reduction, 16,384 sentences x K = 1,000 samples, D = 32, sentences
        sharded over the ranks (fixed total: STRONG scaling), all-gather of the [B_r] results.
The one JSON line also carries the other multi-GPU modes as nested objects (`also_measured`: the other log q(z) mode and
the mmd / iw workloads on the same ranks), and at --gpus 1 the single-GPU time of the 1m workload, so that the 1 -> 8
curve of every fixed-size job is on record whichever workload is the headline.  With --gpus > 1 the line has
`exchange_phases_ms`: per-rank HIP-event times of the phases of an exchange step (own block, time the compute stream
waited for the all-gather, rest block, merge + sums, all-reduce), so that "the all-gather is hidden" is a number.

One "step" = one pass of the hot path over one batch of synthetic latents already resident in HBM:
(N > 1: all-gather of the posterior parameters over RCCL, hidden behind the rank's own block of the pair
matrix) -> operand prep -> fused fp32-MFMA pair log-density + logsumexp -> per-sample log q(z|x), log p(z) ->
float64 sums (N > 1: all-reduce) -> marginal KL and MI.

Prints ONE JSON line (rank 0).  `value` is whole-job pair-evaluations per second.
"""
import argparse
import collections
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
BASE_N = 65536
NORTH_STAR_N = 1048576
MMD_N = 262144                     # BASELINE configs[3]: n1 = n2
IW_B = 16384                       # BASELINE configs[4]: sentences
ALPHAS = [0.1 * i for i in range(15)]      # loss_and_optimisation.py:598


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("auto", "c2", "1m", "weak", "mmd", "iw"), default="auto")
    ap.add_argument("--d", type=int, default=32, help="latent dimension (32: the metric; 64: BASELINE configs[2])")
    ap.add_argument("--n", type=int, default=0, help="override N = M (mmd: n1 = n2; iw: sentences) -- development / tests; "
                                                     "not a BASELINE workload")
    ap.add_argument("--perms", type=int, default=100, help="mmd: permutations (BASELINE configs[3]: 100)")
    ap.add_argument("--k", type=int, default=1000, help="iw: samples per sentence (BASELINE configs[4]: 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the nested secondary measurements")
    return ap.parse_args(argv)


# ---- self-launch: one process per GPU ---------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N fresh child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) and wait for them.  This parent never touches a GPU.  Rank 0's
    stdout is this process's stdout; any rank failing fails the run (the others are stopped by their PIDs)."""
    port = _free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LVAE_BENCH_CHILD="1")
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    alive = set(range(len(procs)))
    while alive:
        for i in sorted(alive):
            r = procs[i].poll()
            if r is None:
                continue
            alive.discard(i)
            if r != 0 and rc == 0:
                rc = r
                print(f"bench.py: rank {i} exited with status {r}; stopping the other ranks", file=sys.stderr, flush=True)
                for j in alive:
                    procs[j].terminate()
        time.sleep(0.05)
    return rc


# ---- workloads ------------------------------------------------------------------------------------------------
def weak_problem_size(world):
    """N = M so that N*M/world == BASE_N^2 (weak scaling), rounded to a multiple of 256*world."""
    q = 256 * world
    return max(q, int(round(BASE_N * math.sqrt(world) / q)) * q)


def resolve_workload(args, world):
    """-> (name, n_total, scaling, description)"""
    name = args.workload
    if name == "auto":
        name = "c2" if world == 1 else "1m"
    if name == "mmd":
        n = ((args.n or MMD_N) // world) * world
        tag = "BASELINE configs[3]" if (n, args.d, args.perms) == (MMD_N, 32, 100) else "development size of configs[3]"
        return name, n, "strong" if world > 1 else "none", (
            f"{tag}: Gaussian-kernel MMD permutation test, n1 = n2 = {n:,}, D = {args.d}, {args.perms} permutations, "
            f"15 bandwidths; rows sharded over {world} GPU(s), fixed total")
    if name == "iw":
        n = ((args.n or IW_B) // world) * world
        tag = "BASELINE configs[4]" if (n, args.d, args.k) == (IW_B, 32, 1000) else "development size of configs[4]"
        return name, n, "strong" if world > 1 else "none", (
            f"{tag}: IW log-likelihood reduction (sampling + log-weight logsumexp), {n:,} sentences x K = {args.k} "
            f"samples, D = {args.d}; sentences sharded over {world} GPU(s), fixed total")
    if args.n:
        n = (args.n // world) * world
        return "custom", n, "strong", f"custom N = M = {n} (development size, not a BASELINE workload)"
    if name == "c2":
        if world != 1:
            raise SystemExit("workload c2 is the single-GPU configuration; use 1m or weak with --gpus > 1")
        return name, BASE_N, "none", ("BASELINE configs[1]: aggregate-posterior log q(z) matrix + logsumexp -> "
                                      "marginal KL / MI, fused fp32 MFMA")
    if name == "1m":
        tag = "BASELINE configs[2]" if args.d == 64 else "north star 'scaling to 8 GPUs at N=1M' (configs[2] shape at the metric's D)"
        return name, NORTH_STAR_N, "strong", (f"{tag}: marginal KL / MI at N = M = 1,048,576, D = {args.d}, rows "
                                              f"sharded over {world} GPU(s), fixed total")
    return name, weak_problem_size(world), "weak", ("BASELINE configs[1] per GPU: N = M = 65,536*sqrt(gpus), every GPU "
                                                    "evaluates 65,536^2 pairs")


def make_inputs(torch, n_total, D, rank, world, device):
    """SURVEY.md section 8(d) synthetic posteriors; every rank draws its own row block (seeded by rank)."""
    n_local = n_total // world
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    mu = torch.randn(n_local, D, device=device, generator=g)
    logvar = 0.5 * torch.randn(n_local, D, device=device, generator=g) - 1.0
    eps = torch.randn(n_local, D, device=device, generator=g)
    z = mu + eps * torch.exp(0.5 * logvar)
    return mu, logvar, z


def cpu_baseline(torch, mu, logvar, z, log_norm, gpu_log_q_z, D, budget_s=12.0, max_rows=4096):
    """Reference op chain (PyTorch eager, CPU) on a bounded row sample of the same workload."""
    from oracle import torch_eager
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    torch.set_num_threads(threads)
    mu_c, lv_c, z_c = mu.cpu(), logvar.cpu(), z[:max_rows].cpu()
    rows, t_used, outs = 0, 0.0, []
    torch_eager.log_qz_rows(mu_c, lv_c, z_c[:8], log_norm)          # warm-up
    while rows < max_rows and t_used < budget_s:
        t0 = time.perf_counter()
        lq, _ = torch_eager.log_qz_rows(mu_c, lv_c, z_c[rows:rows + 64], log_norm)
        t_used += time.perf_counter() - t0
        outs.append(lq)
        rows += 64
    ref = torch.cat(outs).double()
    got = gpu_log_q_z[:rows].double().cpu()
    parity = float((got.mean() - ref.mean()).abs() / ref.mean().abs())
    maxabs = float((got - ref).abs().max())
    pairs = rows * mu_c.shape[0]
    return {"value": pairs / t_used, "unit": "pair-evals/s", "cores": threads, "kind": "port",
            "sample": f"PyTorch-eager restatement of approximate_log_q_z (oracle/torch_eager.py), "
                      f"{rows} of {z.shape[0]} z rows x all {mu_c.shape[0]} posteriors, D={D}, fp32, "
                      f"{t_used:.1f} s",
            "parity_rel_mean_log_q_z": parity, "parity_max_abs": maxabs}


class Bench:
    """One rank of the benchmark."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # LVAE_BENCH_REHEARSAL=1: several ranks on ONE GPU with gloo, to rehearse the multi-rank code path on a
        # single-GPU box.  LVAE_BENCH_KERNELS=<module>: test hook -- the per-rank compute comes from the named module
        # (`local_stats`, `BlockKernels`) and everything runs on the CPU over gloo: the CPU test suite uses it to
        # exercise launch, sharding, exchange and reporting without a GPU.  Numbers from either mean nothing.
        self.kernels_mod = os.environ.get("LVAE_BENCH_KERNELS") or None
        self.rehearsal = os.environ.get("LVAE_BENCH_REHEARSAL") == "1" or self.kernels_mod is not None
        if self.kernels_mod:
            self.device = torch.device("cpu")
        else:
            ndev = torch.cuda.device_count()
            if ndev < 1:
                raise SystemExit("bench.py needs a GPU (there is no CPU path)")
            if not self.rehearsal and local_rank >= ndev:
                raise SystemExit(f"rank {local_rank} has no GPU ({ndev} visible)")
            idx = local_rank % ndev
            torch.cuda.set_device(idx)
            self.device = torch.device("cuda", idx)
        if self.world > 1:
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.device)
        from lvae import distributed as LD
        self.LD = LD
        self.local_stats, self.block_kernels = LD.hip_local_stats, LD.HipBlockKernels
        self.permutation_sums, self.reparam_iw = LD.hip_permutation_sums, LD.hip_reparam_iw
        if self.kernels_mod:
            mod = importlib.import_module(self.kernels_mod)
            self.local_stats, self.block_kernels = mod.local_stats, mod.BlockKernels
            self.permutation_sums, self.reparam_iw = mod.permutation_sums, mod.reparam_iw
        self.gpu = self.device.type == "cuda"

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def measure(self, n_total, D, steps, warmup, ramp):
        """`ramp` + `warmup` untimed steps, then exactly `steps` timed ones between two fences; returns a dict with
        the max-over-ranks elapsed time, the pair kernel's summed HIP-event time and the statistics of the last step."""
        torch, dist, LD, world = self.torch, self.dist, self.LD, self.world
        mu, logvar, z = make_inputs(torch, n_total, D, self.rank, world, self.device)
        log_norm = math.log(n_total)
        # Every step's device results AND the handle of its asynchronous all-reduce are held until the handle has
        # been waited on; a step retires the one from two steps back (long complete: the wait costs nothing and
        # the compute stream never stalls behind the newest collective).
        pending = collections.deque()

        def retire(entry):
            if entry[3] is not None:
                entry[3].wait()
            return entry

        def step():
            if world > 1:
                cur = LD.sharded_latent_sums(mu, logvar, z, method="all", counts=[n_total // world] * world,
                                             local_stats=self.local_stats, block_kernels=self.block_kernels,
                                             async_reduce=True)
            else:
                sums, lq = self.local_stats(mu, logvar, z, mu, logvar, log_norm, False)
                cur = (sums, lq, n_total, None)
            pending.append(cur)
            while len(pending) > 2:
                retire(pending.popleft())

        # Untimed prelude before the W warm-up steps: the GPU clock needs ~8 launches (35 ms) of the 65,536^2 kernel
        # to ramp from idle (4.8 ms for the first launch, 4.04 ms in steady state); reported as "clock_ramp_steps".
        # A failure of the (overlapped) exchange in any step is fatal: the exception ends this rank with a non-zero
        # status and the launcher stops the others -- there is no fallback path.
        for _ in range(ramp + warmup):
            step()
        self.fence()
        prof = None
        if self.gpu:
            import ctypes
            from lvae import _native as N
            N.check(N.lib().lvae_profile_begin(4 * steps))   # (the overlapped multi-GPU step launches the pair kernel twice)
            prof = (ctypes, N)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        elapsed = time.perf_counter() - t0
        last = None
        while pending:
            last = retire(pending.popleft())
        self.sync()
        k_ms, k_n = 0.0, 0
        if prof:
            ctypes, N = prof
            a, b = ctypes.c_double(0), ctypes.c_int(0)
            N.check(N.lib().lvae_profile_end(ctypes.byref(a), ctypes.byref(b)))
            k_ms, k_n = a.value, b.value
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.rehearsal else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        out = LD.stats_from_sums(last[0], last[2], last[1])
        return {"elapsed": elapsed, "k_ms": k_ms, "k_n": k_n, "out": out, "inputs": (mu, logvar, z), "log_norm": log_norm}

    # ---- kernel-time brackets of liblvae (HIP events on the stream the kernels run on) ----------------------------
    def _prof_begin(self, n):
        if not self.gpu:
            return None
        import ctypes
        from lvae import _native as N
        N.check(N.lib().lvae_profile_begin(n))
        return (ctypes, N)

    @staticmethod
    def _prof_end(prof):
        if not prof:
            return 0.0, 0
        ctypes, N = prof
        a, b = ctypes.c_double(0), ctypes.c_int(0)
        N.check(N.lib().lvae_profile_end(ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def _max_over_ranks(self, elapsed):
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def _gather_phases(self, per_step):
        """Mean per-phase milliseconds of this rank's instrumented steps -> (list over ranks, max over ranks)."""
        mine = {}
        for ph in per_step:
            for k, v in ph.items():
                mine[k] = mine.get(k, 0.0) + v / len(per_step)
        ranks = [mine]
        if self.world > 1:
            ranks = [None] * self.world
            self.dist.all_gather_object(ranks, mine)
        worst = {k: max(r.get(k, 0.0) for r in ranks) for k in ranks[0]}
        return ranks, worst

    def exchange_phases(self, n_total, D, steps=3):
        """Instrumented steps of the overlapped exchange, OUTSIDE the timed region (synchronous all-reduce, stamps on the
        compute stream), plus the all-gathers alone: what the collective costs when nothing hides it."""
        torch, dist, LD, world = self.torch, self.dist, self.LD, self.world
        mu, logvar, z = make_inputs(torch, n_total, D, self.rank, world, self.device)
        counts = [n_total // world] * world
        per_step = []
        for i in range(steps + 1):
            tl = LD.Timeline(self.device)
            LD.sharded_latent_sums(mu, logvar, z, method="all", counts=counts, local_stats=self.local_stats,
                                   block_kernels=self.block_kernels, timeline=tl)
            ph = tl.phases_ms()
            if i:                        # (the first one warms the shapes up)
                per_step.append(ph)
        alone = []
        gathered = torch.empty((n_total, D), dtype=mu.dtype, device=mu.device)
        for i in range(steps + 1):
            self.fence()
            tl = LD.Timeline(self.device)
            tl.mark("start")
            for src in (mu, logvar):
                LD._all_gather_into(gathered, src.contiguous(), None)
            tl.mark("all_gather_alone")
            if i:
                alone.append(tl.phases_ms())
        ranks, worst = self._gather_phases(per_step)
        _, worst_alone = self._gather_phases(alone)
        n_r = n_total // world
        return {"steps": steps, "per_rank": ranks, "max_over_ranks": worst,
                "all_gather_alone_ms": worst_alone.get("all_gather_alone"),
                "bytes": {"all_gather_received_per_rank": 2 * n_total * D * 4,
                          "all_gather_sent_per_rank": 2 * n_r * D * 4, "all_reduce": 5 * 8},
                "note": "HIP events on the compute stream (rehearsal: host clock); wait_all_gather = time the compute "
                        "stream sat behind the two all-gathers after its own block was done (hidden when ~0)"}

    # ---- BASELINE configs[3]: MMD permutation test ---------------------------------------------------------------
    def measure_mmd(self, n_each, D, P, steps, warmup, phase_steps=0):
        torch, LD, world, rank = self.torch, self.LD, self.world, self.rank
        from lvae import estimators as E
        n_r = n_each // world
        g = torch.Generator(device=self.device).manual_seed(4321 + rank)
        x = torch.randn(n_r, D, device=self.device, generator=g)
        y = 0.98 * torch.randn(n_r, D, device=self.device, generator=g) + 0.01
        # the labelings must be the same on every rank: one seed, drawn by every rank for itself (label generation is
        # part of a user's call -- timed separately below -- but not of the kernel-sum step)
        gl = torch.Generator(device=self.device).manual_seed(99)
        self.sync()
        t0 = time.perf_counter()
        labels = E.permutation_labels(n_each, n_each, P, self.device, gl)
        self.sync()
        t_labels = time.perf_counter() - t0
        counts = ([n_r] * world, [n_r] * world)

        def step(tl=None):
            if world > 1:
                return LD.sharded_mmd_permutation_test(x, y, ALPHAS, labels, group=None, counts=counts,
                                                       permutation_sums=self.permutation_sums, timeline=tl)
            pooled = torch.cat((x, y), 0).contiguous()
            q, lr, tot = self.permutation_sums(pooled, 0, pooled, labels, ALPHAS, 1e-5, True)
            stats = E.permutation_statistics(q, lr, tot, n_each, n_each)
            return {"statistic": stats[0], "statistics": stats[1:], "pval": (stats[1:] >= stats[0]).double().mean()}

        for _ in range(warmup):
            step()
        self.fence()
        prof = self._prof_begin(4 * steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.fence()
        elapsed = self._max_over_ranks(time.perf_counter() - t0)
        k_ms, k_n = self._prof_end(prof)
        phases = None
        if phase_steps and world > 1:
            per = []
            for _ in range(phase_steps):
                tl = LD.Timeline(self.device)
                step(tl)
                per.append(tl.phases_ms())
            ranks, worst = self._gather_phases(per)
            phases = {"steps": phase_steps, "per_rank": ranks, "max_over_ranks": worst,
                      "bytes": {"all_gather_received_per_rank": 2 * n_each * D * 4, "all_reduce": (2 * (P + 1) + 1) * 8}}
        return {"elapsed": elapsed, "k_ms": k_ms, "k_n": k_n, "labels_s": t_labels, "phases": phases,
                "statistic": float(out["statistic"]), "pval": float(out["pval"])}

    def mmd_record(self, m, n_each, D, P, steps):
        n = 2.0 * n_each
        entries = n * n
        # SURVEY 8(d) prices an evaluated entry at 2 D flop (distance) + 2 (P + 1) flop (Y = K S, observed labeling
        # included) and the symmetric matrix at half its entries, against the fp32 MFMA peak
        flop_rank = 0.5 * entries * (2.0 * D + 2.0 * (P + 1)) / self.world
        k_avg = m["k_ms"] / steps if m["k_n"] else None
        achieved = flop_rank / (k_avg * 1e-3) / 1e12 if k_avg else None
        rec = {"value": entries * steps / m["elapsed"], "unit": "pooled-entries/s", "ms_per_step": m["elapsed"] / steps * 1e3,
               "steps": steps, "n1": n_each, "n2": n_each, "D": D, "permutations": P,
               "ms_label_generation": m["labels_s"] * 1e3,
               "ms_per_step_with_label_generation": (m["elapsed"] / steps + m["labels_s"]) * 1e3,
               "statistic": m["statistic"], "pval": m["pval"],
               "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": achieved / FP32_MFMA_PEAK_TFLOPS if achieved else None, "traffic": None,
                            "kernel": "pair_lse_mfma_kernel<.., PERM> (LVAE_PATH_AUTO: split-bf16 distance + bf16 label "
                                      "MFMAs; priced with SURVEY 8(d)'s fp32 flop count, so > 1 is possible)",
                            "kernel_ms_avg": k_avg, "launches": m["k_n"], "flops_per_step_per_rank": flop_rank},
               "note": "parity unpinned (torch-two-sample absent)"}
        if m["phases"]:
            rec["exchange_phases_ms"] = m["phases"]
        return rec

    # ---- BASELINE configs[4]: IW log-likelihood reduction ----------------------------------------------------------
    def measure_iw(self, B, K, D, steps, warmup, phase_steps=0):
        torch, LD, world, rank = self.torch, self.LD, self.world, self.rank
        B_r = B // world
        g = torch.Generator(device=self.device).manual_seed(777 + rank)
        mu = torch.randn(B_r, D, device=self.device, generator=g)
        lv = 0.5 * torch.randn(B_r, D, device=self.device, generator=g) - 1.0
        eps = torch.randn(B_r, K, D, device=self.device, generator=g)
        lpxz = 5.0 * torch.randn(B_r, K, device=self.device, generator=g) - 100.0
        counts = [B_r] * world

        def step(tl=None):
            if world > 1:
                return LD.sharded_iw_log_p_x(mu, lv, eps, lpxz, counts=counts, reparam_iw=self.reparam_iw, timeline=tl)
            return self.reparam_iw(mu, lv, eps, lpxz)

        for _ in range(warmup):
            step()
        self.fence()
        prof = self._prof_begin(2 * steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.fence()
        elapsed = self._max_over_ranks(time.perf_counter() - t0)
        k_ms, k_n = self._prof_end(prof)
        phases = None
        if phase_steps and world > 1:
            per = []
            for _ in range(phase_steps):
                tl = LD.Timeline(self.device)
                step(tl)
                per.append(tl.phases_ms())
            ranks, worst = self._gather_phases(per)
            phases = {"steps": phase_steps, "per_rank": ranks, "max_over_ranks": worst,
                      "bytes": {"all_gather_received_per_rank": B * 4}}
        return {"elapsed": elapsed, "k_ms": k_ms, "k_n": k_n, "phases": phases,
                "iw_ll_mean": float(out.double().mean()), "rows_out": int(out.numel())}

    def iw_record(self, m, B, K, D, steps):
        units = float(B) * K
        bytes_rank = units * (4 * D + 4) / self.world          # eps + log p(x|z) read per (sentence, sample)
        k_avg = m["k_ms"] / steps if m["k_n"] else None
        achieved = bytes_rank / (k_avg * 1e-3) / 1e9 if k_avg else None
        rec = {"value": units * steps / m["elapsed"], "unit": "(sentence,sample)/s", "ms_per_step": m["elapsed"] / steps * 1e3,
               "steps": steps, "B": B, "K": K, "D": D, "iw_ll_mean": m["iw_ll_mean"], "gathered_rows": m["rows_out"],
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": None,
                            "kernel": "sample_iw_kernel (reduction only: eps and log p(x|z) in, iw_ll out)",
                            "kernel_ms_avg": k_avg, "launches": m["k_n"], "bytes_per_step_per_rank": bytes_rank}}
        if m["phases"]:
            rec["exchange_phases_ms"] = m["phases"]
        return rec

    def summary(self, m, n_total, D, steps):
        """value / ms_per_step / roofline numbers of one measurement."""
        pairs_total = float(n_total) * float(n_total)
        pairs_rank = pairs_total / self.world
        k_avg_ms = m["k_ms"] / steps
        achieved = 4.0 * D * pairs_rank / (k_avg_ms * 1e-3) / 1e12 if m["k_n"] else None
        value = pairs_total * steps / m["elapsed"]
        return {"value": value, "ms_per_step": m["elapsed"] / steps * 1e3, "pairs_rank": pairs_rank,
                "k_avg_ms": k_avg_ms, "achieved": achieved,
                "pct_of_peak": value * 4 * D / (self.world * FP32_MFMA_PEAK_TFLOPS * 1e12) * 100.0}

    def side_configs(self, D):
        """BASELINE configs[3] and configs[4] on these ranks, timed after the headline measurement and nested in the same
        line (side records: the headline `value` never depends on them; an error is recorded as text).  Collective:
        every rank takes part."""
        out = {}
        fail = 0
        for key, fn in (("configs4_iw", lambda: self.iw_record(self.measure_iw(IW_B, 1000, D, 20, 2, 2), IW_B, 1000, D, 20)),
                        ("configs3_mmd_perm", lambda: self.mmd_record(self.measure_mmd(MMD_N, D, 100, 2, 1, 1), MMD_N, D, 100, 2))):
            try:
                out[key] = fn()
            except Exception as e:                 # noqa: BLE001 -- a side record must never take the headline down
                out[key] = {"error": repr(e)}
                fail = 1
            if self.world > 1:                     # a rank that failed alone would leave the others inside a collective:
                t = self.torch.tensor([fail], dtype=self.torch.int32, device="cpu" if self.rehearsal else self.device)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                if int(t.item()):                  # stop the side runs on every rank together
                    out.setdefault(key, {"error": "another rank failed"})
                    break
        if self.world == 1 and "configs4_iw" in out and "error" not in out["configs4_iw"]:
            # one GPU: the form that also writes the two [B,K] density matrices, next to the reduction-only one
            try:
                from lvae import estimators as E
                torch = self.torch
                g = torch.Generator(device=self.device).manual_seed(7)
                mu = torch.randn(IW_B, D, device=self.device, generator=g)
                lv = 0.5 * torch.randn(IW_B, D, device=self.device, generator=g) - 1.0
                eps = torch.randn(IW_B, 1000, D, device=self.device, generator=g)
                lpxz = 5.0 * torch.randn(IW_B, 1000, device=self.device, generator=g) - 100.0
                for _ in range(3):
                    E.reparam_iw(mu, lv, eps, lpxz, return_z=False)
                self.sync()
                prof = self._prof_begin(40)
                for _ in range(20):
                    E.reparam_iw(mu, lv, eps, lpxz, return_z=False)
                self.sync()
                k_ms, k_n = self._prof_end(prof)
                by = IW_B * 1000 * (4 * D + 4 + 8)
                out["configs4_iw"]["with_density_matrices"] = {
                    "kernel_ms_avg": k_ms / max(k_n, 1), "GBps": by / (k_ms / max(k_n, 1) * 1e-3) / 1e9,
                    "frac_of_8TBps": by / (k_ms / max(k_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "bytes_per_unit": 4 * D + 4 + 8}
            except Exception as e:                 # noqa: BLE001
                out["configs4_iw"]["with_density_matrices"] = {"error": repr(e)}
        return out

    def run_side_headline(self, name, n_total, scaling, desc):
        """`--workload mmd` / `--workload iw`: BASELINE configs[3] / configs[4] as the headline of the line."""
        args, world, rank = self.args, self.world, self.rank
        D = args.d
        if name == "mmd":
            m = self.measure_mmd(n_total, D, args.perms, args.steps, args.warmup, 2)
            rec = self.mmd_record(m, n_total, D, args.perms, args.steps)
            metric = "MMD permutation test: pooled kernel-matrix entries/sec"
            cfg = {"workload": desc, "name": name, "n1": n_total, "n2": n_total, "D": D, "permutations": args.perms,
                   "rows_per_gpu": n_total // world,
                   "parallelism": f"rows of both samples sharded x{world}, all-gather(x, y) + all-reduce(2P+1 sums)",
                   "exchange": ("all-gather of both samples = %.1f MB received per rank; all-reduce of %d float64"
                                % (2 * n_total * D * 4 / 1e6, 2 * (args.perms + 1) + 1)) if world > 1 else "none"}
        else:
            m = self.measure_iw(n_total, args.k, D, args.steps, args.warmup, 2)
            rec = self.iw_record(m, n_total, args.k, D, args.steps)
            metric = "IW log-likelihood reduction: (sentence, sample)-evals/sec"
            cfg = {"workload": desc, "name": name, "B": n_total, "K": args.k, "D": D, "sentences_per_gpu": n_total // world,
                   "parallelism": f"sentences sharded x{world}, all-gather of the [B_r] results",
                   "exchange": ("all-gather of %d float32 = %.1f KB received per rank" % (n_total, n_total * 4 / 1e3))
                   if world > 1 else "none"}
        if rank == 0:
            line = {"metric": metric, "value": rec["value"], "unit": rec["unit"], "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": scaling,
                    "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg, "roofline": rec["roofline"],
                    "results": {k: rec[k] for k in ("statistic", "pval", "iw_ll_mean", "gathered_rows",
                                                    "ms_label_generation", "ms_per_step_with_label_generation") if k in rec}}
            if "exchange_phases_ms" in rec:
                line["exchange_phases_ms"] = rec["exchange_phases_ms"]
            if self.rehearsal:
                line["rehearsal"] = "several ranks on one device / injected per-rank compute: the numbers mean nothing"
            print(json.dumps(line), flush=True)
        if world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()

    def run(self):
        args, world, rank, torch = self.args, self.world, self.rank, self.torch
        D = args.d
        name, n_total, scaling, desc = resolve_workload(args, world)
        if name in ("mmd", "iw"):
            return self.run_side_headline(name, n_total, scaling, desc)
        long_step = float(n_total) ** 2 / world > 4.0 * BASE_N ** 2
        ramp = 2 if long_step else 10
        m = self.measure(n_total, D, args.steps, args.warmup, ramp)
        s = self.summary(m, n_total, D, args.steps)
        line = None
        if rank == 0:
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and world == 1 and name == "c2" and D == 32:
                with open(tpath) as f:
                    traffic = json.load(f).get("pair_lse_mfma_kernel_hbm_bytes_per_launch")
            n_r = n_total // world
            line = {
                "metric": "log q(z|x) pair-evals/sec at D=32" if D == 32 else f"log q(z|x) pair-evals/sec at D={D}",
                "value": s["value"], "unit": "pair-evals/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_steps": ramp,
                "ms_per_step": s["ms_per_step"],
                "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": desc, "name": name,
                           "N": n_total, "M": n_total, "D": D, "pairs_per_gpu": s["pairs_rank"],
                           "rows_per_gpu": n_r,
                           "parallelism": f"row-sharded x{world}, all-gather(mu,logvar)+all-reduce(sums)",
                           "exchange": ("two all-gathers (mu, logvar) of %d x [%d, %d] fp32 each = %.1f MB received per "
                                        "rank, overlapped with the own-posterior block; async all-reduce of 5 float64 "
                                        "sums (40 B)" % (world, n_r, D, 2 * world * n_r * D * 4 / 1e6)) if world > 1 else "none"},
                "pct_of_fp32_mfma_peak": s["pct_of_peak"],
                "roofline": {"bound": "mfma", "achieved": s["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s",
                             "frac": (s["achieved"] / FP32_MFMA_PEAK_TFLOPS) if s["achieved"] else None,
                             "traffic": traffic,
                             "kernel": "pair_lse_mfma_kernel (fp32 MFMA contraction, fixed-reference logsumexp epilogue)",
                             "kernel_ms_avg": s["k_avg_ms"], "launches": m["k_n"],
                             "launches_per_step": m["k_n"] / args.steps,
                             "flops_per_launch": 4.0 * D * s["pairs_rank"] * args.steps / max(m["k_n"], 1)},
                "result_read": "device-resident sums, read by the host once after the timed steps",
                "results": {k: v for k, v in m["out"].items() if isinstance(v, float)},
            }
            if self.rehearsal:
                line["rehearsal"] = "several ranks on one device / injected per-rank compute: the numbers mean nothing"
        # ---- secondary measurements, nested in the same line (every rank takes part in the collective ones) ----
        if not args.no_side_runs and not args.n:
            side = {}
            if world > 1:
                other = "weak" if name == "1m" else "1m"
                n2 = weak_problem_size(world) if other == "weak" else NORTH_STAR_N
                st2 = max(3, min(args.steps, 20 if other == "weak" else 5))
                m2 = self.measure(n2, D, st2, 2, 2)
                s2 = self.summary(m2, n2, D, st2)
                side[other] = {"N": n2, "value": s2["value"], "unit": "pair-evals/s", "ms_per_step": s2["ms_per_step"],
                               "steps": st2, "scaling": "weak" if other == "weak" else "strong",
                               "pct_of_fp32_mfma_peak": s2["pct_of_peak"], "kernel_ms_avg": s2["k_avg_ms"]}
                del m2
                # the other sharded configurations on the same ranks, so that every one of them gets its curve from
                # the driver's default runs (errors are recorded as text, the headline never depends on them)
                side.update(self.side_configs(D))
            elif name == "c2" and self.gpu:
                # the fixed-size multi-GPU job on ONE GPU: the base point of its strong-scaling curve
                st2 = 2
                m2 = self.measure(NORTH_STAR_N, D, st2, 1, 0)
                s2 = self.summary(m2, NORTH_STAR_N, D, st2)
                side["1m"] = {"N": NORTH_STAR_N, "value": s2["value"], "unit": "pair-evals/s",
                              "ms_per_step": s2["ms_per_step"], "steps": st2, "scaling": "strong",
                              "pct_of_fp32_mfma_peak": s2["pct_of_peak"], "kernel_ms_avg": s2["k_avg_ms"],
                              "marginal_kl": m2["out"]["marginal_kl"], "mi": m2["out"]["mi"]}
                del m2
            if rank == 0:
                line["also_measured"] = side
        if world > 1 and not args.no_side_runs:
            ph = self.exchange_phases(n_total, D)
            if rank == 0:
                line["exchange_phases_ms"] = ph
        if rank == 0 and world == 1 and self.gpu and name == "c2":
            from lvae import _native as N
            mu, logvar, z = m["inputs"]
            log_norm = m["log_norm"]
            pairs_total = float(n_total) ** 2
            # Not the headline: the same step with the pair kernel swapped for the opt-in split-bf16 one
            # (LVAE_PATH_MFMA_BF16X3: three bf16 limbs per fp32 value on the bf16 matrix pipe, fp32-equivalent
            # results), timed after the headline measurement.  `value` above stays the fp32 MFMA path.
            def step_bx():
                sums, _ = self.LD.hip_local_stats(mu, logvar, z, mu, logvar, log_norm, False, path=N.PATH_MFMA_BF16X3)
                return sums                   # read after the loop, like the headline step
            try:
                for _ in range(2):
                    step_bx()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    sums = step_bx()
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                s_lq, _, s_lp, _, cnt = sums.tolist()
                line["experimental_bf16x3"] = {
                    "value": pairs_total * args.steps / el, "unit": "pair-evals/s",
                    "ms_per_step": el / args.steps * 1e3, "marginal_kl": (s_lq - s_lp) / cnt,
                    "rel_diff_marginal_kl_vs_fp32_path": abs((s_lq - s_lp) / cnt - m["out"]["marginal_kl"])
                    / abs(m["out"]["marginal_kl"]),
                    "note": "opt-in path, not the north star's fp32 MFMA contraction; see DESIGN.md"}
            except Exception as e:                       # never lose the headline line over the side measurement
                line["experimental_bf16x3"] = {"error": repr(e)}
            if not args.no_side_runs and D == 32:
                line["also_measured"].update(self.side_configs(D))
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(torch, mu, logvar, z, log_norm, m["out"]["log_q_z_local"], D)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    Bench(args).run()


if __name__ == "__main__":
    main()
