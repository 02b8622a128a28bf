"""approximatenn_amd -- MI355X-native (gfx950, HIP) backend for approximateNN's precomp()/query() hot path.

The product is the C-ABI shared library under csrc/ (include/*.h); this package is the thin host-side
mirror of the reference's interface used by the tests, the bench and multi-GPU hosts.
"""
from . import _lib
from .api import HostStream, Index, Save, checksum, gpu_cleanup, gpu_init, precomp, query, recall_ranks, recall_summary, synth_randnorm

__all__ = ["HostStream", "Index", "Save", "precomp", "query", "recall_ranks", "recall_summary", "gpu_init", "gpu_cleanup", "synth_randnorm", "checksum",
           "_lib"]
