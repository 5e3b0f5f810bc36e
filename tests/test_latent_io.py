"""The reference's on-disk latent formats: round trip, safe loading, and --
on the GPU -- the CLI against the oracle."""
import json
import os
import pickle

import numpy as np
import pytest
import torch


def _enc(n=96, D=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(n, D, generator=g)
    lv = 0.5 * torch.randn(n, D, generator=g) - 1
    z = mu + torch.randn(n, D, generator=g) * torch.exp(0.5 * lv)
    return z, mu, lv


def test_round_trip_reference_style_pickle(tmp_path):
    from lvae import latent_io
    z, mu, lv = _enc()
    # written exactly like utils_train.analyse_save_latents does: pickle.dump of a dict of CPU tensors
    # plus NumPy float scalars
    enc = dict(latents=z, mus=mu, logvars=lv, stds=lv.mul(0.5).exp(), log_q_z_x=torch.zeros(len(z)),
               log_p_z=torch.zeros(len(z)), elbo=np.mean([1.0, 2.0]), tts_mmd_loss=np.float64(0.25))
    p = tmp_path / "encodings-epoch-1-global-step-10.p"
    pickle.dump(enc, open(p, "wb"))
    got = latent_io.load_encodings(p)
    for k in latent_io.ENCODING_KEYS:
        assert torch.equal(got[k], enc[k])
    assert float(got["elbo"]) == 1.5
    p2 = tmp_path / "x.p"
    latent_io.save_encodings(p2, z, mu, lv)
    again = pickle.load(open(p2, "rb"))          # our writer is readable by the reference's plain pickle.load
    assert torch.equal(again["latents"], z) and torch.allclose(again["stds"], torch.exp(0.5 * lv))
    np.savez(tmp_path / "e.npz", latents=z.numpy(), mus=mu.numpy(), logvars=lv.numpy())
    assert torch.equal(latent_io.load_encodings(tmp_path / "e.npz")["mus"], mu)
    torch.save(dict(latents=z, mus=mu, logvars=lv), tmp_path / "e.pt")
    assert torch.equal(latent_io.load_encodings(tmp_path / "e.pt")["logvars"], lv)


def test_malicious_pickle_is_refused(tmp_path):
    from lvae import latent_io

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))

    p = tmp_path / "evil.p"
    pickle.dump({"latents": Evil()}, open(p, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        latent_io.load_tensor_dict(p)


def test_bad_encodings_rejected(tmp_path):
    from lvae import latent_io
    z, mu, lv = _enc()
    p = tmp_path / "bad.p"
    pickle.dump(dict(latents=z, mus=mu[:-1], logvars=lv), open(p, "wb"))
    with pytest.raises(ValueError):
        latent_io.load_encodings(p)
    pickle.dump(dict(foo=z), open(p, "wb"))
    with pytest.raises(ValueError):
        latent_io.load_encodings(p)


@pytest.mark.gpu
def test_cli_analyse_matches_oracle(tmp_path, capsys):
    from lvae import cli, latent_io
    from oracle import oracle_np as onp
    z, mu, lv = _enc(n=640, D=32, seed=3)
    p = tmp_path / "encodings-epoch-0-global-step-1.p"
    latent_io.save_encodings(p, z, mu, lv)
    out = cli.main(["analyse", str(p), "--method", "chen", "--dataset-size", "42068", "--prod-marginals",
                    "--mmd", "--mi-bounds", "128"])
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["n"] == 640
    lq, pm = onp.approximate_log_q_z(mu.numpy(), lv.numpy(), z.numpy(), "chen", 42068, True, False)
    lp = onp.sample_log_likelihood(z.numpy(), reduce_batch_dim=False)
    lqx = onp.sample_log_likelihood(z.numpy(), mu.numpy(), lv.numpy(), reduce_batch_dim=False)
    want = {"marginal_kl": (lq - lp).mean(), "mi": (lqx - lq).mean(), "tc": (lq - pm).mean(),
            "dim_kl": (pm - lp).mean(), "kl_analytical": onp.kl_divergence(mu.numpy(), lv.numpy(), 0.5)[0]}
    for k, v in want.items():
        assert abs(out[k] - v) <= 1e-4 * abs(v) + 1e-5, (k, out[k], v)
    lows, ups = zip(*[onp.calc_upper_and_lower_bound_representational_mi(
        z[b:b + 128].numpy(), np.concatenate([mu[b:b + 128].numpy(), lv[b:b + 128].numpy()], 1))
        for b in range(0, 640, 128)])
    assert abs(out["mi_lower_infonce"] - np.mean(lows)) <= 1e-4 * abs(np.mean(lows))
    assert abs(out["mi_upper_mbu"] - np.mean(ups)) <= 1e-4 * abs(np.mean(ups))
    assert out["mi_bound_batches"] == 5 and "tts_mmd" in out and "mmd_biased" in out


@pytest.mark.gpu
def test_cli_iw_reports_ks_distance(tmp_path, capsys):
    """The IW result dict of dist_iw_log_likelihood_x_obs_x_gen.py:184-198 -> means + the KS distance between the
    observed and generated IW-LL vectors (utils_train.py:496-499)."""
    from scipy import stats
    from lvae import cli
    rng = np.random.default_rng(2)
    d = {k: torch.from_numpy((rng.standard_normal(n) * 15 - 90 + off).astype(np.float32))
         for k, n, off in (("log_p_x_obs_valid", 3370, 0.0), ("log_p_x_w_obs_valid", 3370, 0.0),
                           ("log_p_x_gen", 2000, 4.0), ("log_p_x_w_gen", 2000, 1.0))}
    d["lens_gen"] = torch.arange(2000)
    p = tmp_path / "iw_result.p"
    pickle.dump(d, open(p, "wb"))
    out = cli.main(["iw", str(p)])
    capsys.readouterr()
    for kind in ("log_p_x", "log_p_x_w"):
        want = stats.ks_2samp(d[f"{kind}_obs_valid"].numpy(), d[f"{kind}_gen"].numpy(), method="asymp").statistic
        assert out[f"D_ks_{kind}_obs_valid_vs_gen"] == want
    assert abs(out["log_p_x_gen_mean"] - float(d["log_p_x_gen"].double().mean())) < 1e-9
