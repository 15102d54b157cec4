"""multiz_amd -- MI355X (gfx950) implementation of the multiz block-pair merge DP
(yama()/pre_yama(), reference mz_yama.c / mz_preyama.c / mz_scores.c) behind a C ABI.

The product is libmzamd.so (multiz_amd/csrc: hand-written HIP kernels + a C host shim);
this package is only the Python binding used by the tests and bench.py.
"""
from .api import (lib, build, yama_batch, yama_one, set_scores_hoxd70, set_scores_hoxd85,  # noqa: F401
                  MZ_STATUS, DevBatch, LIB_PATH, preyama_batch)
