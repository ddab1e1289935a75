"""bpldenoising_amd -- MI355X-native inner TV-denoising solver behind BPLDenoising's
evaluate/solve surface (tv_op_learning_function / denoise / TVDenoise).

Product path: libbpltv.so (hand-written HIP for gfx950, C ABI in include/bpltv.h).  This package
is the thin host mirror of the reference's operator interface; it has no CPU fallback.
"""
from .learning_function import (FwdGradientOp, L2CostFunction, TVDenoise, TVSolver, denoise,
                                generate_cost, tv_op_learning_function, sumregs_learning_function, sumregs_denoise,
                                use_devices)
from .sharding import ShardedLearningFunction, shard_range
from .datasets import testdataset, load_filelist_dataset
from . import trbox
from . import experiments
from .experiments import (scalar_bilevel_tv_learn, patch_bilevel_tv_learn, scalar_bilevel_sumregs_learn,
                          patch_bilevel_sumregs_learn, validate_sumregs_parameter)

__all__ = ["FwdGradientOp", "L2CostFunction", "TVDenoise", "TVSolver", "denoise",
           "tv_op_learning_function", "sumregs_learning_function", "sumregs_denoise", "generate_cost", "use_devices", "ShardedLearningFunction", "shard_range", "testdataset",
           "load_filelist_dataset", "trbox", "experiments", "scalar_bilevel_tv_learn", "patch_bilevel_tv_learn",
           "scalar_bilevel_sumregs_learn", "patch_bilevel_sumregs_learn", "validate_sumregs_parameter"]
