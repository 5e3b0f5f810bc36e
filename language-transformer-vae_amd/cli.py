"""``python -m lvae.cli`` -- latent-analysis statistics straight from the reference's on-disk dumps.

  python -m lvae.cli analyse encodings-epoch-3-global-step-9000.p [--method chen --dataset-size 42068]
                             [--prod-marginals] [--mmd] [--mi-bounds 128] [--device cuda:0]
  python -m lvae.cli iw iw_results.pickle

``analyse`` evaluates, on the GPU, what the reference computes from such a dump by hand (notebooks) or
per training step (loss_and_optimisation.py:816-824): marginal KL, MI, TC, dimension-wise KL, analytic
KL, MMD against the prior and the InfoNCE / MBU MI bounds, and prints one JSON object.
"""
import argparse
import json
import sys

import torch


def analyse(args):
    from . import estimators as E
    from . import latent_io
    from . import mutual_information as MI
    from .distributed import hip_local_stats
    import math
    enc = latent_io.load_encodings(args.file)
    dev = torch.device(args.device)
    z, mu, lv = (enc[k].float().to(dev).contiguous() for k in ("latents", "mus", "logvars"))
    n, D = z.shape
    if args.method == "chen":
        if args.dataset_size is None:
            raise SystemExit("--method chen needs --dataset-size")
        log_norm = math.log(n * args.dataset_size)
    else:
        log_norm = math.log(n)
    sums, _ = hip_local_stats(mu, lv, z, mu, lv, log_norm, args.prod_marginals)
    s_lq, s_pm, s_lp, s_lqx, cnt = sums.tolist()
    out = {"file": args.file, "n": n, "latent_dim": D, "method": args.method,
           "log_q_z": s_lq / cnt, "log_p_z": s_lp / cnt, "log_q_z_x": s_lqx / cnt,
           "marginal_kl": (s_lq - s_lp) / cnt, "mi": (s_lqx - s_lq) / cnt}
    if args.prod_marginals:
        out.update(log_q_z_prod_marg=s_pm / cnt, tc=(s_lq - s_pm) / cnt, dim_kl=(s_pm - s_lp) / cnt)
    kl, fb = E.kl_divergence(mu, lv, hinge_kl_loss_lambda=args.free_bits)
    out.update(kl_analytical=float(kl), fb_kl_analytical=float(fb))
    if args.mmd:
        g = torch.Generator(device=dev).manual_seed(args.seed)
        prior = torch.randn(z.shape, device=dev, generator=g)
        out["mmd_biased"] = float(E.maximum_mean_discrepancy(z, prior))
        out["tts_mmd"] = float(E.get_tts_mmmd(z.unsqueeze(1), prior))
    if args.mi_bounds:
        K = args.mi_bounds
        lows, ups = [], []
        for b0 in range(0, n - K + 1, K):       # batches of K like calc_all_mi_bounds (mutual_information.py:281-356)
            lo, up = MI.calc_upper_and_lower_bound_representational_mi(
                z[b0:b0 + K], torch.cat([mu[b0:b0 + K], lv[b0:b0 + K]], dim=1))
            lows.append(lo)
            ups.append(up)
        if lows:
            out.update(mi_lower_infonce=sum(lows) / len(lows), mi_upper_mbu=sum(ups) / len(ups),
                       mi_bound_batches=len(lows), mi_bound_batch_size=K)
    print(json.dumps(out))
    return out


def iw(args):
    from . import latent_io
    d = latent_io.load_tensor_dict(args.file)
    out = {"file": args.file}
    for k, v in d.items():
        if isinstance(v, torch.Tensor) and v.numel() > 0 and v.is_floating_point():
            out[k + "_mean"] = float(v.double().mean())
            out[k + "_n"] = int(v.numel())
    # KS distance between the observed-data and the generated-data IW-LL vectors (the "-D_ks" metric of
    # utils_train.py:496-499), for every pair the dump holds
    if torch.cuda.is_available():
        from . import pareto
        for obs in ("obs_valid", "obs_train"):
            for kind in ("log_p_x", "log_p_x_w"):
                a, b = d.get(f"{kind}_{obs}"), d.get(f"{kind}_gen")
                if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.numel() and b.numel():
                    out[f"D_ks_{kind}_{obs}_vs_gen"] = pareto.ks_2samp(a.to(args.device), b.to(args.device))[0]
    print(json.dumps(out))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="lvae")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("analyse", help="marginal KL / MI / TC / MMD / MI bounds from an encodings dump")
    a.add_argument("file")
    a.add_argument("--method", choices=["all", "chen"], default="all")
    a.add_argument("--dataset-size", type=int, default=None)
    a.add_argument("--prod-marginals", action="store_true")
    a.add_argument("--mmd", action="store_true")
    a.add_argument("--mi-bounds", type=int, default=0, metavar="K")
    a.add_argument("--free-bits", type=float, default=0.5)
    a.add_argument("--seed", type=int, default=0)
    a.add_argument("--device", default="cuda:0")
    a.set_defaults(fn=analyse)
    i = sub.add_parser("iw", help="summarise an importance-weighted log-likelihood result dict")
    i.add_argument("file")
    i.add_argument("--device", default="cuda:0")
    i.set_defaults(fn=iw)
    args = ap.parse_args(argv)
    return args.fn(args)


if __name__ == "__main__":
    main(sys.argv[1:])
