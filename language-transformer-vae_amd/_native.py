"""ctypes binding of liblvae.so -- the C ABI declared in include/lvae.h.

There is no fallback: if the shared library is missing or a call fails, this raises.  PyTorch is used
only to own device memory and to name the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVAE_LIB", os.path.join(_HERE, "liblvae.so"))

SPREAD_LOGVAR, SPREAD_VAR = 0, 1
PATH_AUTO, PATH_MFMA, PATH_DIRECT = 0, 1, 2
PATH_MFMA_BF16X3 = 3      # split-bf16 contraction (include/lvae.h): opt-in for log q(z), AUTO's choice for large kernel sums

_lib = None

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float

_SIGNATURES = {
    "lvae_last_error": (ctypes.c_char_p, []),
    "lvae_version": (_int, []),
    "lvae_log_qz_workspace_bytes": (_int, [_i64, _i64, _int, _int, ctypes.POINTER(_i64)]),
    "lvae_log_qz_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _int, _i64, _i64, _int, _f32, _int, _i64, _int,
                               _c_f32p, _c_f32p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_log_qz_noalloc_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _int, _i64, _i64, _int, _f32, _int, _i64, _int,
                                       _c_f32p, _c_f32p, ctypes.c_void_p]),
    "lvae_log_qz_partial_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _int, _i64, _i64, _int, _int,
                                       _c_f32p, _c_f32p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_qz_estimate_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _int, _i64, _i64, _int, _f32, _c_f32p,
                                    ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_sample_ll_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p,
                                  ctypes.c_void_p]),
    "lvae_sample_ll_elem_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _c_f32p,
                                       ctypes.c_void_p]),
    "lvae_reparam_iw_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p,
                                   _c_f32p, _c_f32p, ctypes.c_void_p]),
    "lvae_encode_tail_supported": (_int, [_i64, _i64, _int]),
    "lvae_encode_tail_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _c_f32p, _c_f32p,
                                    _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "lvae_encode_tail_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _c_f32p, _c_f32p, _c_f32p,
                                        _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "lvae_encode_tail_ws_supported": (_int, [_i64, _i64, _int]),
    "lvae_encode_tail_workspace_bytes": (_int, [_i64, _i64, _int, ctypes.POINTER(_i64)]),
    "lvae_encode_tail_ws_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _c_f32p, _c_f32p,
                                       _c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, ctypes.c_void_p, _i64,
                                       ctypes.c_void_p]),
    "lvae_encode_tail_bwd_ws_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _c_f32p, _c_f32p,
                                           _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, ctypes.c_void_p, _i64,
                                           ctypes.c_void_p]),
    "lvae_iw_lse_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _c_f32p, ctypes.c_void_p]),
    "lvae_reduce_f32": (_int, [_c_f32p, _i64, _int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                               _i64, ctypes.c_void_p]),
    "lvae_latent_sums_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, ctypes.c_void_p,
                                    ctypes.c_void_p]),
    "lvae_latent_stats_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _int, _c_f32p, _c_f32p,
                                     _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_pair_bounds_workspace_bytes": (_int, [_i64, _int, ctypes.POINTER(_i64)]),
    "lvae_pair_bounds_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _int, _int, ctypes.c_void_p,
                                    ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_categorical_bounds_workspace_bytes": (_int, [_i64, ctypes.POINTER(_i64)]),
    "lvae_categorical_bounds_f32": (_int, [_c_f32p, ctypes.c_void_p, _i64, _i64, ctypes.c_void_p, ctypes.c_void_p,
                                           _i64, ctypes.c_void_p]),
    "lvae_ks_2samp_workspace_bytes": (_int, [_i64, _i64, ctypes.POINTER(_i64)]),
    "lvae_ks_2samp_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, ctypes.c_void_p, ctypes.c_void_p, _i64,
                                 ctypes.c_void_p]),
    "lvae_lse_combine_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _f32, _c_f32p, _c_f32p, _c_f32p,
                                    ctypes.c_void_p]),
    "lvae_lse_rebase_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, ctypes.c_void_p]),
    "lvae_log_prob_pairs_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _int, _i64, _i64, _int, _c_f32p,
                                       ctypes.c_void_p]),
    "lvae_gauss_ll_masked_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, ctypes.POINTER(_i64),
                                        ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                        ctypes.POINTER(_i64), _int, _c_f32p, ctypes.c_void_p]),
    "lvae_kl_diag_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, _f32, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                ctypes.c_void_p]),
    "lvae_colmean_f32": (_int, [_c_f32p, _i64, _i64, _c_f32p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_kde_workspace_bytes": (_int, [_i64, _i64, ctypes.POINTER(_i64)]),
    "lvae_kde_score_f32": (_int, [_c_f32p, _c_f32p, _i64, _i64, _i64, _int, _f32, _int, _c_f32p, ctypes.c_void_p, _i64,
                                  ctypes.c_void_p]),
    "lvae_mmd_workspace_bytes": (_int, [_i64, _i64, _int, ctypes.POINTER(_i64)]),
    "lvae_mmd_rbf_f32": (_int, [_c_f32p, _c_f32p, _i64, _i64, _int, ctypes.POINTER(_f32), _int, _f32, _int, _int,
                                ctypes.c_void_p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_kernel_sum_workspace_bytes": (_int, [_i64, _i64, _int, ctypes.POINTER(_i64)]),
    "lvae_kernel_sum_f32": (_int, [_c_f32p, _c_f32p, _i64, _i64, _int, ctypes.POINTER(_f32), _int, _f32, _int,
                                   _i64, _int, _int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _i64,
                                   ctypes.c_void_p]),
    "lvae_log_qz_bwd_workspace_bytes": (_int, [_i64, _i64, _int, _int, ctypes.POINTER(_i64)]),
    "lvae_log_qz_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _f32, _c_f32p, _c_f32p, _c_f32p,
                                   _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_sample_ll_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _c_f32p, _int, _c_f32p,
                                      _c_f32p, _c_f32p, ctypes.c_void_p]),
    "lvae_reparam_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p,
                                    ctypes.c_void_p]),
    "lvae_kl_diag_bwd_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, _f32, _c_f32p, _c_f32p, _int, _c_f32p,
                                    _c_f32p, ctypes.c_void_p]),
    "lvae_plan_info": (_int, [_i64, _i64, _int, _int, ctypes.POINTER(_i64)]),
    "lvae_layout_info": (_int, [_int, _i64, _i64, _int, _int, ctypes.POINTER(_i64)]),
    "lvae_mmd_perm_workspace_bytes": (_int, [_i64, _i64, _int, ctypes.POINTER(_i64)]),
    "lvae_mmd_perm_f32": (_int, [_c_f32p, _i64, _i64, _c_f32p, _i64, _int, ctypes.c_void_p, _int,
                                 ctypes.POINTER(_f32), _int, _f32, _int, _int, ctypes.c_void_p, ctypes.c_void_p, _i64,
                                 ctypes.c_void_p]),
    "lvae_kernel_matrix_f32": (_int, [_c_f32p, _c_f32p, _i64, _i64, _int, ctypes.POINTER(_f32), _int, _f32,
                                      _c_f32p, ctypes.c_void_p, _i64, ctypes.c_void_p]),
    "lvae_profile_begin": (_int, [_int]),
    "lvae_profile_end": (_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_int)]),
    "lvae_dev_reload_env": (_int, []),
}

NOALLOC_SCRATCH_BYTES = 64 << 20
REDUCE_WS_BYTES = 8192
KMAT_WS_BYTES = 4096


class LvaeError(RuntimeError):
    pass


def lib():
    """Load liblvae.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvaeError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}/csrc` "
                            "(there is no CPU or PyTorch fallback)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc):
    if rc != 0:
        raise LvaeError(f"liblvae status {rc}: {lib().lvae_last_error().decode()}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def as_f32(t, name):
    """The reference works in float32 on one device; keep that contract explicit."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.requires_grad:
        # under torch.no_grad() the reference's plain torch ops just evaluate; with grad mode on, an estimator
        # that reaches this point has no backward behind it (the differentiable ones detach first)
        if torch.is_grad_enabled():
            raise LvaeError(f"{name} requires grad: this estimator is forward-only (no HIP backward behind it)")
        t = t.detach()
    if not t.is_cuda:
        raise LvaeError(f"{name} is on {t.device}: the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def workspace(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def reduce_f64(x, mode=0, scale=1.0):
    """scale * sum(x) (mode 0) or scale * sum(exp(x)) (mode 1) as a 0-dim float64 device tensor."""
    x = as_f32(x, "x").reshape(-1)
    dev = x.device
    with torch.cuda.device(dev):
        out = torch.empty((), dtype=torch.float64, device=dev)
        ws = workspace(REDUCE_WS_BYTES, dev)
        check(lib().lvae_reduce_f32(ptr(x), x.numel(), mode, float(scale), ptr(out), ptr(ws),
                                    REDUCE_WS_BYTES, stream_ptr(dev)))
    return out


def mean_f32(x):
    """Mean of a float32 device vector as a 0-dim float32 tensor (accumulated in float64 on device)."""
    return reduce_f64(x, 0, 1.0 / x.numel()).float()


def colmean_f32(x):
    """Mean over dim 0 of a [n, ...] float32 device tensor -> [...]."""
    x = as_f32(x, "x")
    n = x.shape[0]
    flat = x.reshape(n, -1)
    c = flat.shape[1]
    dev = x.device
    with torch.cuda.device(dev):
        out = torch.empty(c, dtype=torch.float32, device=dev)
        ws = workspace(8 * c + 256, dev)
        check(lib().lvae_colmean_f32(ptr(flat), n, c, ptr(out), ptr(ws), 8 * c + 256, stream_ptr(dev)))
    return out.reshape(x.shape[1:])


def host_floats(values):
    arr = (ctypes.c_float * len(values))(*[float(v) for v in values])
    return arr, len(values)


def reload_env():
    """Have the library read its LVAE_* environment switches again (they are read once per process); for sweeps and
    tests that change one inside a running process."""
    check(lib().lvae_dev_reload_env())
