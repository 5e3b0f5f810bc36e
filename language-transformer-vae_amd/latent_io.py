"""On-disk latent formats of the reference, read without executing arbitrary pickle code.

* encodings dump of ``utils_train.analyse_save_latents`` (NewsVAE/utils_train.py:552-571): a pickled dict
  ``{latents, mus, logvars, stds, log_q_z_x, log_p_z}`` of CPU tensors (+ a few float scalars),
  file name ``encodings-epoch-<e>-global-step-<s>.p``;
* IW result dict of ``dist_iw_log_likelihood_x_obs_x_gen`` (NewsVAE/iw_log_likelihood_x_obs_x_gen/
  dist_iw_log_likelihood_x_obs_x_gen.py:184-198): ``{log_p_x_obs_valid, log_p_x_w_obs_valid, lens_obs_valid,
  ..._train, log_p_x_gen, log_p_x_w_gen, lens_gen}`` of CPU tensors.

Both are plain ``pickle.dump`` files.  They are read with an unpickler that only admits the handful of
globals a dict of tensors / NumPy scalars needs; anything else raises.  ``.pt`` (``torch.save``) and
``.npz`` files with the same keys are accepted too.
"""
import io
import pickle

import numpy as np
import torch

ENCODING_KEYS = ("latents", "mus", "logvars", "stds", "log_q_z_x", "log_p_z")

_ALLOWED = {
    ("torch._utils", "_rebuild_tensor_v2"),
    ("torch._utils", "_rebuild_tensor"),
    ("torch.storage", "_load_from_bytes"),
    ("collections", "OrderedDict"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
    ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"),
}
_ALLOWED_TORCH_TYPES = {"FloatStorage", "DoubleStorage", "LongStorage", "IntStorage", "HalfStorage",
                        "BFloat16Storage", "BoolStorage", "float32", "float64", "int64", "int32"}


def _safe_load_from_bytes(b):
    return torch.load(io.BytesIO(b), weights_only=True)


class _TensorDictUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _safe_load_from_bytes          # storages are re-read with torch's weights-only loader
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and name in _ALLOWED_TORCH_TYPES:
            return getattr(torch, name)
        raise pickle.UnpicklingError(f"refusing to load global {module}.{name}: not part of a tensor dict")


def load_tensor_dict(path):
    """Read a dict of tensors / scalars from a reference-style pickle, a ``.pt`` or an ``.npz`` file."""
    p = str(path)
    if p.endswith(".npz"):
        with np.load(p, allow_pickle=False) as f:
            return {k: torch.from_numpy(np.asarray(f[k])) for k in f.files}
    if p.endswith(".pt") or p.endswith(".pth"):
        obj = torch.load(p, weights_only=True, map_location="cpu")
    else:
        with open(p, "rb") as f:
            obj = _TensorDictUnpickler(f).load()
    if not isinstance(obj, dict):
        raise ValueError(f"{p}: expected a dict, found {type(obj).__name__}")
    return obj


def load_encodings(path):
    """The encodings dict of analyse_save_latents (utils_train.py:552-559); checks keys and shapes."""
    d = load_tensor_dict(path)
    missing = [k for k in ("latents", "mus", "logvars") if k not in d]
    if missing:
        raise ValueError(f"{path}: not an encodings file, missing {missing}")
    lat, mu, lv = d["latents"], d["mus"], d["logvars"]
    if lat.dim() != 2 or lat.shape != mu.shape or lat.shape != lv.shape:
        raise ValueError(f"{path}: latents/mus/logvars must share one [n, D] shape")
    return d


def save_encodings(path, latents, mus, logvars, log_q_z_x=None, log_p_z=None, **extra):
    """Write the same dict the reference writes (utils_train.py:552-571): CPU tensors, ``pickle.dump``."""
    enc = dict(latents=latents.cpu(), mus=mus.cpu(), logvars=logvars.cpu(),
               stds=logvars.cpu().mul(0.5).exp_())
    if log_q_z_x is not None:
        enc["log_q_z_x"] = log_q_z_x.cpu()
    if log_p_z is not None:
        enc["log_p_z"] = log_p_z.cpu()
    enc.update(extra)
    with open(path, "wb") as f:
        pickle.dump(enc, f)
    return enc
