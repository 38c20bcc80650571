"""MI355X-native seed-and-extend overlap engine for PacBio long reads (hot path of vmingchen/PacBioAssembly).

The product is pacbioassembly_amd/lib/libpba.so (hand-written HIP for gfx950 behind the C ABI in
include/pba.h) plus the C++ compat headers in include/compat/.  `engine` wraps the C ABI for pytest and
bench.py.  There is no CPU implementation in this package.
"""
from . import _lib, engine  # noqa: F401
from .engine import Consensus, Context, PbaError, ProbeTable, SeedIndex, SeqSet  # noqa: F401

__all__ = ["Consensus", "Context", "PbaError", "ProbeTable", "SeedIndex", "SeqSet", "engine"]
