"""blu_amd -- MI355X (gfx950) implementation of the factorize hot path of the `blu` crate (rwl/blu).

Product surface: include/blu_hip.h (C ABI, libblu_hip.so) and `blu_amd.BLU`, the host-side mirror of
the reference's `struct BLU`.  Nothing here imports the CPU oracle (oracle/): that is test
infrastructure.
"""
from . import keys  # noqa: F401
from . import blu  # noqa: F401
from .blu import BLU, BluError, SELFCHECK_LIB_PATH, build_library, factorize_batch, gen_lp_basis, lib  # noqa: F401
from .maxvolume import maxvolume  # noqa: F401
