"""carmel_amd — MI355X-native EM (forward-backward) and Gibbs training for carmel WFST cascades.

The compute path is the C-ABI library `libcarmel_hip.so` (include/carmel_hip.h; sources in carmel_amd/csrc):
hand-written gfx950 HIP kernels plus the host-side lattice builder.  This package holds the thin Python mirror of
the reference's training interface used by the tests and by bench.py, and the synthetic-workload generators.
"""
from ._capi import LIB_PATH, CarmelHipError, get_option, lib, option_names, options_from_env, set_option  # noqa: F401  (fails loudly when the library is missing)
from .model import Corpus, Wfst  # noqa: F401
from .trainer import HipForwardBackward, TrainOpts, train  # noqa: F401
