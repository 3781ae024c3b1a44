"""ddcmd_amd -- MI355X-native Martini MD inner loop behind ddcMD's plugin surface.

The product is the C-ABI shared library ``libddcmi.so`` (host C + hand-written HIP
kernels for gfx950, see include/ddcmi.h).  This Python package is only the thin
loader/driver used by tests and bench.py; it never computes forces itself and
raises if the HIP library is missing.
"""
from ._lib import load_library, LibraryMissing  # noqa: F401
from .deck import Setup, load_deck, units_convert  # noqa: F401
from .synth import make_water_setup, replicate_setup  # noqa: F401
