"""rayuela.jl_amd -- MI355X (gfx950) PQ/OPQ encode + ADC linear scan behind Rayuela.jl's API.

Host-side mirror of the reference's operator interface for this path (same names, argument order,
defaults and index bases as src/PQ.jl, src/OPQ.jl, src/Linscan.jl), implemented as thin calls into
the C ABI of librayuela_hip.so (include/rayuela_hip.h).  The Julia drop-in files that bind the same
ABI with `ccall` live in julia/.  No CPU fallback exists: without the HIP library every call raises.
"""
from ._lib import RayuelaHipError, lib, lib_path, set_tuning, last_timing  # noqa: F401
from .utils import splitarray, cat_codebooks  # noqa: F401
from .xvecs import fvecs_read, ivecs_read, bvecs_read, fvecs_write, ivecs_write  # noqa: F401
from .PQ import quantize_pq, quantize_pq_u8  # noqa: F401
from .OPQ import quantize_opq, rotate  # noqa: F401
from .RVQ import quantize_rvq, quantize_rvq_u8  # noqa: F401


from .PQ import train_pq, kmpp_seeds  # noqa: F401,E402
from .OPQ import train_opq  # noqa: F401,E402
from .RVQ import train_rvq  # noqa: F401,E402
from .Linscan import (linscan_pq, linscan_opq, linscan_lsq, linscan_cq, linscan_aqd_query, LsqIndex,  # noqa: F401
                      linscan_aqd_query_extra_byte, eval_recall)

from .index import Index, Dataset  # noqa: F401,E402
from . import h5results  # noqa: F401,E402  (libhdf5 is looked up lazily, on first use)
from . import datasets  # noqa: F401,E402

__all__ = ["quantize_pq", "quantize_opq", "linscan_pq", "linscan_opq", "linscan_lsq", "linscan_cq",
           "eval_recall", "splitarray"]
