"""Host mirror of the one helper of src/utils.jl that the hot path uses."""
import numpy as np


def splitarray(x, nparts):
    """src/utils.jl:179-203.  Splits `x` (a range / sequence) in `nparts` contiguous parts; if
    len(x) is not a multiple of nparts the first len(x) % nparts parts carry one extra element."""
    x = list(x) if not isinstance(x, range) else x
    n = len(x)
    per, extra = divmod(n, nparts)
    out, pos = [], 0
    for i in range(nparts):
        size = per + (1 if i < extra else 0)
        out.append(x[pos:pos + size])
        pos += size
    return out


def _as_f32(a, name):
    a = np.asarray(a)
    if a.dtype != np.float32:
        # src/PQ.jl:32 allocates costs as Float32, so the reference only dispatches for Float32 data
        raise TypeError("%s must be float32 (the reference encode only dispatches for Float32)" % name)
    return np.ascontiguousarray(a)


def cat_codebooks(C):
    """Vector{Matrix} -> one flat buffer: concatenation of the m [h][sub_i] blocks
    (== cat(C..., dims=3) of src/Linscan.jl:22 when all sub_i are equal)."""
    return np.concatenate([_as_f32(c, "C[i]").reshape(-1) for c in C])
