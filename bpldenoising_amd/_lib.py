"""ctypes binding of libbpltv.so (the C ABI of include/bpltv.h).

The library is the product; there is no CPU fallback.  If the shared object is missing or cannot
be loaded this module raises -- it never routes to another implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPLTV_LIB_PATH", os.path.join(_HERE, "libbpltv.so"))  # override: A/B builds

_dp = C.POINTER(C.c_double)


class BpltvParams(C.Structure):
    """struct bpltv_params (include/bpltv.h); defaults from
    /root/reference/src/TVLearningFunctionVec.jl:33-43 via bpltv_default_params."""
    _fields_ = [
        ("rho", C.c_double), ("tau0", C.c_double), ("sigma0", C.c_double),
        ("accel", C.c_int), ("maxiter", C.c_int),
        ("delta_t", C.c_double),
        ("check_every", C.c_int),
        ("gap_tol", C.c_double),
        ("tile_iters", C.c_int), ("use_graph", C.c_int),
        ("kappa_cap", C.c_double),
        ("refine", C.c_int),
        ("deterministic", C.c_int),
        ("reserved", C.c_int * 5),
        ("init", C.c_int), ("order", C.c_int),
        ("opnorm", C.c_double),
    ]


class BpltvStats(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("O", C.c_int), ("device", C.c_int),
        ("iterations", C.c_int), ("launches", C.c_int), ("tile_iters", C.c_int), ("tiles", C.c_int),
        ("region_i", C.c_int), ("region_j", C.c_int),
        ("graph_used", C.c_int),
        ("pdhg_ms", C.c_double), ("cost_ms", C.c_double), ("adjoint_ms", C.c_double), ("total_ms", C.c_double),
        ("bytes_per_px_iter", C.c_double), ("algorithmic_bytes", C.c_double),
        ("last_gap", C.c_double), ("adjoint_residual", C.c_double), ("adjoint_residual_raw", C.c_double),
        ("kappa_used", C.c_double),
        ("adjoint_attempts", C.c_int), ("adjoint_method", C.c_int),
        ("reg_gradient_used", C.c_int), ("ngpus", C.c_int),
        ("shards", C.c_int), ("collective", C.c_int),
        ("collective_ms", C.c_double),
        ("nccl_ranks", C.c_int), ("hb_sync", C.c_int), ("adjoint_chunks", C.c_int),
        ("pdhg_variant", C.c_int),
        ("ncu", C.c_int), ("launch_chains", C.c_int), ("sweep_shards", C.c_int), ("reserved_i", C.c_int),
        ("launch_host_ms", C.c_double * 2),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("reserved", "reserved_i", "launch_host_ms")}
        d["launch_host_ms"] = [self.launch_host_ms[0], self.launch_host_ms[1]]
        d["adjoint_method"] = {1: "band", 2: "bcr", 3: "band-hbm", 4: "band-lu", 5: "nd", 6: "nd-lu"}.get(self.adjoint_method, "")
        d["hb_sync"] = {0: "", 1: "event", 2: "value"}.get(self.hb_sync, "")
        d["collective"] = {0: "none", 1: "ncclAllReduce", 2: "ncclAllGather+ordered sum", 3: "host sum"}.get(self.collective, "")
        return d


# every symbol include/bpltv.h declares: name -> (restype, argtypes)
_H = C.c_void_p
_PP = C.POINTER(BpltvParams)
SYMBOLS = {
    "bpltv_version": (C.c_int, []),
    "bpltv_default_params": (C.c_int, [_PP]),
    "bpltv_create": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bpltv_create_multi": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bpltv_create_sharded": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]),
    "bpltv_destroy": (C.c_int, [_H]),
    "bpltv_set_data": (C.c_int, [_H, _dp, _dp]),
    "bpltv_set_data_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "bpltv_denoise": (C.c_int, [_H, _dp, C.c_int, C.c_int, _PP, _dp]),
    "bpltv_denoise_device": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, _PP]),
    "bpltv_evaluate": (C.c_int, [_H, _dp, C.c_int, C.c_int, C.c_double, _PP, _dp, _dp, _dp]),
    "bpltv_sumregs_default_params": (C.c_int, [_PP]),
    "bpltv_sumregs_denoise": (C.c_int, [_H, _dp, C.c_int, C.c_int, _PP, _dp]),
    "bpltv_sumregs_evaluate": (C.c_int, [_H, _dp, C.c_int, C.c_int, C.c_double, _PP, _dp, _dp, _dp]),
    "bpltv_evaluate_partial": (C.c_int, [_H, _dp, C.c_int, C.c_int, C.c_double, _PP, _dp, _dp]),
    "bpltv_evaluate_device": (C.c_int, [_H, _dp, C.c_int, C.c_int, C.c_double, _PP, C.c_void_p]),
    "bpltv_u_device": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "bpltv_copy_u_device": (C.c_int, [_H, C.c_void_p]),
    "bpltv_duality_gap": (C.c_int, [_H, _dp]),
    "bpltv_grad_fwd": (C.c_int, [_H, _dp, _dp, _dp]),
    "bpltv_grad_fwd_adjoint": (C.c_int, [_H, _dp, _dp, _dp]),
    "bpltv_gradient": (C.c_int, [_H, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _PP, _dp]),
    "bpltv_sweep": (C.c_int, [_H, _dp, C.c_int, C.c_int, C.c_int, _PP, _dp, _dp]),
    "bpltv_per_image": (C.c_int, [_H, _dp]),
    "bpltv_set_option": (C.c_int, [_H, C.c_char_p, C.c_double]),
    "bpltv_stats": (C.c_int, [_H, C.POINTER(BpltvStats)]),
    "bpltv_last_error": (C.c_char_p, [_H]),
}

_lib = None


class BpltvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbpltv error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libbpltv.so and bind every declared symbol.  Raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bpldenoising_amd: %s is missing -- the HIP library has not been built "
            "(run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root). "
            "There is no CPU fallback." % LIB_PATH)
    # Coexistence with PyTorch-ROCm in one process: torch ships its own libamdhip64/libhsa-runtime.
    # Whichever HIP runtime is loaded first serves both (same SONAME); two initialised runtimes in
    # one process leave the second without GPUs.  So when torch is installed, let it load first.
    if os.environ.get("BPLTV_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
