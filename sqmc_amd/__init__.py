"""sqmc_amd -- MI355X (gfx950) implementation of sqmc's semistochastic walker step,
deterministic-core matvec and HCI connection generation behind a C ABI
(include/sqmc_gpu.h, built into sqmc_amd/libsqmc_gpu.so).

The Python in this package is host-side plumbing only (ctypes door, input tables, the
scalar population-control logic of the reference's walk loop).  There is no CPU fallback:
every compute entry point fails loudly when the HIP library or a GPU is missing.
"""
from ._lib import set_device, load_library, build_library, GpuChem, SpmvPlan, SqmcGpuError, RNG_REPLAY, RNG_COUNTER  # noqa: F401
