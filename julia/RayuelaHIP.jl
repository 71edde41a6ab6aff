# RayuelaHIP.jl -- drop-in bodies for Rayuela.jl's PQ/OPQ encode and ADC linear scan on MI355X.
#
# Host code stays Julia; every body is a `ccall` into librayuela_hip.so (include/rayuela_hip.h).
# Signatures, defaults, return types and index bases are those of the reference:
#   quantize_pq   src/PQ.jl:18-48        quantize_opq  src/OPQ.jl:19-27
#   linscan_pq    src/Linscan.jl:5-37    linscan_opq   src/Linscan.jl:93-115
#   linscan_lsq   src/Linscan.jl:118-157 linscan_cq    src/Linscan.jl:160-193   (SURVEY 8f rank 2)
#   quantize_rvq  src/RVQ.jl:18-66                                              (SURVEY 8f rank 3)
# Julia's column-major arrays are passed as they are: a d-by-n Matrix{Float32} is the C array
# [n][d] the library expects, an m-by-n Matrix{UInt8} is [n][m], k-by-nq outputs are [nq][k].
#
# NOTE: this image has no Julia toolchain, so this file has not been executed here; the identical
# C ABI is exercised through ctypes by tests/ (rayuela.jl_amd/*.py mirrors this file line by line).
module RayuelaHIP

# only quantize_rvq's singleton re-pick needs them (both are Rayuela.jl dependencies, Manifest.toml:79-83,135-139)
import Clustering, Distances

export quantize_pq, quantize_opq, quantize_rvq, linscan_pq, linscan_opq, linscan_lsq, linscan_cq, train_pq, train_opq, train_rvq
export HipIndex, set_codes!, set_codes_synth!, search, HipDataset, quantize

# Multi-GPU without touching a call site: with ENV["RAYUELA_HIP_DEVICES"] = "0,1,2,3" (or "all") set before the
# call, linscan_pq / linscan_opq below shard B row-wise over those devices inside the library (per-device scan,
# RCCL gather of the per-shard top-k keys, merge on the first device) and quantize_pq / quantize_opq split X over
# them (one PCIe link each).  The results are bit-identical to the single-GPU ones.

# deps/build.jl:64-67 writes the library paths into deps/deps.jl; here one constant / env var.
const librayuela_hip = get(ENV, "RAYUELA_HIP_LIB",
                           joinpath(@__DIR__, "..", "rayuela.jl_amd", "librayuela_hip.so"))

function _check(status::Cint)
  if status != 0
    msg = unsafe_string(ccall((:rq_last_error, librayuela_hip), Cstring, ()))
    error("librayuela_hip status $status: $msg")
  end
  nothing
end

# Result arrays of the scans (the reference's `zeros(Cfloat, k, nq)`, src/Linscan.jl:16-17): every element is written
# by the library, and large ones come page-locked from the library's pool (rq_host_alloc) -- the copy back into a fresh
# pageable array runs at the speed of its first-touch page faults (4.4 ms for the 80 MB of a SIFT1M-shape answer).
# They are ordinary `Matrix{T}` to the caller; the finalizer hands the buffer back when the array is collected.
function _result(::Type{T}, k::Int, nq::Int) where T
  bytes = sizeof(T) * k * nq
  if bytes >= (4 << 20)
    p = ccall((:rq_host_alloc, librayuela_hip), Ptr{Cvoid}, (Csize_t,), bytes)
    if p != C_NULL
      A = unsafe_wrap(Array, Ptr{T}(p), (k, nq); own=false)
      finalizer(a -> ccall((:rq_host_free, librayuela_hip), Cvoid, (Ptr{Cvoid},), pointer(a)), A)
      return A
    end
  end
  return Matrix{T}(undef, k, nq)
end

# cat(C..., dims=3) for even splits; plain concatenation of the column-major blocks otherwise
_cat_codebooks(C::Vector{Matrix{Float32}}) = vcat([vec(Ci) for Ci in C]...)

"""
    quantize_pq(X, C, V=false) -> B     (src/PQ.jl:18-48)
`B::Matrix{Int16}`, m-by-n, one-based.
"""
function quantize_pq(X::Matrix{Float32}, C::Vector{Matrix{Float32}}, V::Bool=false)
  d, n = size(X)
  m    = length(C)
  h    = size(C[1], 2)
  B    = _result(Int16, m, n)
  if V print("Encoding on $m codebooks with librayuela_hip... ") end
  _check(ccall((:rq_encode_pq_i16, librayuela_hip), Cint,
    (Ptr{Int16}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint, Cint),
    B, X, _cat_codebooks(C), Int64(n), Cint(d), Cint(m), Cint(h)))
  if V println("done") end
  return B
end

"""
    quantize_opq(X, R, C, V=false) -> B     (src/OPQ.jl:19-27) == quantize_pq(R' * X, C, V)
"""
function quantize_opq(X::Matrix{Float32}, R::Matrix{Float32}, C::Vector{Matrix{Float32}}, V::Bool=false)
  d, n = size(X)
  m    = length(C)
  h    = size(C[1], 2)
  B    = _result(Int16, m, n)
  _check(ccall((:rq_encode_opq_i16, librayuela_hip), Cint,
    (Ptr{Int16}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint, Cint),
    B, X, R, _cat_codebooks(C), Int64(n), Cint(d), Cint(m), Cint(h)))
  return B
end

"""
    quantize_rvq(X, C, V=false) -> B, singletons     (src/RVQ.jl:18-66)
`B::Matrix{Int16}` m-by-n one-based; `singletons[i]` holds re-picked entries for the unused centres of
codebook i.  The m encode stages and residual updates run on the device; the library returns the
per-centre counts, and only when some are zero is the reference's own randomised re-pick
(`Clustering.repick_unused_centers`, :50-53; needs `using Clustering, Distances` like src/Rayuela.jl)
replayed here on the residual of that stage.
"""
function quantize_rvq(X::Matrix{Float32}, C::Vector{Matrix{Float32}}, V::Bool=false)
  d, n = size(X)
  m    = length(C)
  h    = size(C[1], 2)
  B      = Matrix{Int16}(undef, m, n)
  counts = Matrix{UInt32}(undef, h, m)          # C view [m][h]
  _check(ccall((:rq_encode_rvq_i16, librayuela_hip), Cint,
    (Ptr{Int16}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint, Cint, Ptr{UInt32}, Ptr{Cfloat}),
    B, X, hcat(C...), Int64(n), Cint(d), Cint(m), Cint(h), counts, C_NULL))
  singletons = Vector{Matrix{Float32}}(undef, m)
  if any(counts .== 0)
    Xr = copy(X)
    for i = 1:m
      unused = findall(counts[:, i] .== 0)
      picked = C[i][:, B[i, :]]
      if !isempty(unused)
        costs = vec(sum((Xr .- picked) .^ 2, dims=1))
        temp_codebook = similar(C[i])
        Clustering.repick_unused_centers(Xr, costs, temp_codebook, unused, Distances.SqEuclidean())
        singletons[i] = temp_codebook[:, unused]
      end
      Xr .-= picked
    end
  end
  return B, singletons
end

"""
    train_rvq(X, m, h, niter=25, V=false; seed=0) -> C, B, error     (src/RVQ.jl:86-127)
One k-means per stage on the running residual, on the device, seeded by kmeans++ like the reference's
`kmeans(Xr, h, init=:kmpp, maxiter=niter)` -- with the library's seeded stream instead of Julia's RNG, so runs
agree in objective, not bit for bit (and are bit-reproducible for a given `seed`).
"""
function train_rvq(X::Matrix{Float32}, m::Integer, h::Integer, niter::Integer=25, V::Bool=false; seed::Integer=0)
  d, n = size(X)
  Ccat = Array{Float32}(undef, d, h, m)            # C view [m][h][d]
  B    = _result(Int16, m, n)
  err  = Ref{Cdouble}(0.0)
  _check(ccall((:rq_train_rvq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Int16}, Ref{Cdouble}, Ptr{Cfloat}, Int64, Cint, Cint, Cint, Cint, UInt64),
    Ccat, B, err, X, Int64(n), Cint(d), Cint(m), Cint(h), Cint(niter), UInt64(seed)))
  C = [Ccat[:, :, i] for i = 1:m]
  return C, B, Float32(err[])
end

"""
    linscan_pq(B, X, C, b, k=10000) -> dists, idx     (src/Linscan.jl:5-26)
`B::Matrix{UInt8}` zero-based m-by-n; returns k-by-nq `dists::Matrix{Cfloat}` (ascending) and
`idx::Matrix{Cuint}` ONE-based (the reference's `res .+= 1` is folded into the kernel: id_base = 1).
"""
function linscan_pq(B::Matrix{UInt8}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, b::Int, k::Int=10000)
  m, n  = size(B)
  d, nq = size(X)
  @show k, nq
  dists = _result(Cfloat, k, nq)
  res   = _result(Cuint,  k, nq)
  _check(ccall((:rq_linscan_pq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cuchar}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Cint, Cint, Cint, Cint),
    dists, res, B, cat(C..., dims=3), X, Int64(n), Int64(nq), Cint(m), Cint(d), Cint(k), Cint(1)))
  return dists, res
end

function linscan_pq(B::Matrix{T}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, b::Int, k::Int=10000) where T <: Integer
  B_uint8 = convert(Matrix{UInt8}, B .- 1)      # src/Linscan.jl:35
  return linscan_pq(B_uint8, X, C, b, k)
end

"""
    linscan_opq(B, X, C, b, R, k=10000)     (src/Linscan.jl:93-115) == linscan_pq(B, R' * X, C, b, k)
"""
function linscan_opq(B::Matrix{UInt8}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, b::Int,
                     R::Matrix{Cfloat}, k::Int=10000)
  m, n  = size(B)
  d, nq = size(X)
  dists = _result(Cfloat, k, nq)
  res   = _result(Cuint,  k, nq)
  _check(ccall((:rq_linscan_opq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cuchar}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Cint, Cint, Cint, Cint),
    dists, res, B, cat(C..., dims=3), X, R, Int64(n), Int64(nq), Cint(m), Cint(d), Cint(k), Cint(1)))
  return dists, res
end

function linscan_opq(B::Matrix{T}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, b::Int,
                     R::Matrix{Cfloat}, k::Int=10000) where T <: Integer
  B_uint8 = convert(Matrix{UInt8}, B .- 1)
  return linscan_opq(B_uint8, X, C, b, R, k)
end

"""
    linscan_lsq(B, X, C, dbnorms, R, k=10000) -> dists, idx     (src/Linscan.jl:118-157)
ADC search for additive quantizers, database norms passed apart; `idx` is ONE-based as in the reference.
"""
function linscan_lsq(B::Matrix{UInt8}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}},
                     dbnorms::Vector{Cfloat}, R::Matrix{Cfloat}, k::Int=10000)
  m, n  = size(B)
  d, nq = size(X)
  _, h  = size(C[1])
  dists = _result(Cfloat, k, nq)
  res   = _result(Cuint,  k, nq)
  _check(ccall((:rq_linscan_lsq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cuchar}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat},
     Int64, Int64, Cint, Cint, Cint, Cint, Cint),
    dists, res, B, X, hcat(C...), dbnorms, R, Int64(n), Int64(nq), Cint(m), Cint(h), Cint(d), Cint(k), Cint(1)))
  return dists, res
end

function linscan_lsq(B::Matrix{T}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}},
                     dbnorms::Vector{Cfloat}, R::Matrix{Cfloat}, k::Int=10000) where T <: Integer
  return linscan_lsq(convert(Matrix{UInt8}, B .- 1), X, C, dbnorms, R, k)
end

"""
    HipLsqIndex(B, C, dbnorms)  /  search(ix, X, R, k=10000) -> dists, idx
linscan_lsq over a PREPARED base (rq_lsq_prepare / rq_lsq_search / rq_lsq_release): codes, dbnorms and codebooks stay on the
device together with the pre-filter's O(n) pass over the base, which `linscan_lsq` repeats on every call.  `search` returns
exactly what `linscan_lsq(B, X, C, dbnorms, R, k)` returns (one-based `idx`).
"""
mutable struct HipLsqIndex
  handle::Ptr{Cvoid}
  n::Int
  m::Int
  d::Int
  function HipLsqIndex(B::Matrix{UInt8}, C::Vector{Matrix{Cfloat}}, dbnorms::Vector{Cfloat})
    m, n = size(B)
    d, h = size(C[1])
    hd = ccall((:rq_lsq_prepare, librayuela_hip), Ptr{Cvoid},
               (Ptr{Cuchar}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint, Cint),
               B, hcat(C...), dbnorms, Int64(n), Cint(m), Cint(h), Cint(d))
    hd == C_NULL && error("rq_lsq_prepare: " * unsafe_string(ccall((:rq_last_error, librayuela_hip), Cstring, ())))
    ix = new(hd, n, m, d)
    finalizer(x -> (x.handle != C_NULL && ccall((:rq_lsq_release, librayuela_hip), Cvoid, (Ptr{Cvoid},), x.handle); x.handle = C_NULL), ix)
    return ix
  end
end
HipLsqIndex(B::Matrix{T}, C::Vector{Matrix{Cfloat}}, dbnorms::Vector{Cfloat}) where T <: Integer =
  HipLsqIndex(convert(Matrix{UInt8}, B .- 1), C, dbnorms)

function search(ix::HipLsqIndex, X::Matrix{Cfloat}, R::Matrix{Cfloat}, k::Int=10000)
  d, nq = size(X)
  dists = _result(Cfloat, k, nq)
  res   = _result(Cuint,  k, nq)
  _check(ccall((:rq_lsq_search, librayuela_hip), Cint,
    (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint),
    ix.handle, dists, res, X, R, Int64(nq), Cint(k), Cint(1)))
  return dists, res
end

"""
    linscan_cq(B, X, C, k=10000) -> dists, idx     (src/Linscan.jl:160-193)
"""
function linscan_cq(B::Matrix{UInt8}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, k::Int=10000)
  m, n  = size(B)
  d, nq = size(X)
  _, h  = size(C[1])
  dists = _result(Cfloat, k, nq)
  res   = _result(Cuint,  k, nq)
  _check(ccall((:rq_linscan_cq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cuchar}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Cint, Cint, Cint, Cint, Cint),
    dists, res, B, X, hcat(C...), Int64(n), Int64(nq), Cint(m), Cint(h), Cint(d), Cint(k), Cint(1)))
  return dists, res
end

function linscan_cq(B::Matrix{T}, X::Matrix{Cfloat}, C::Vector{Matrix{Cfloat}}, k::Int=10000) where T <: Integer
  return linscan_cq(convert(Matrix{UInt8}, B .- 1), X, C, k)
end

# splitarray(1:d, m) sizes (src/utils.jl:179-203), to cut the flat codebook buffer back into matrices
function _split_codebooks(Ccat::Vector{Float32}, d::Int, m::Int, h::Int)
  per, extra = divrem(d, m)
  C = Vector{Matrix{Float32}}(undef, m)
  pos = 0
  for i = 1:m
    sub  = per + (i <= extra ? 1 : 0)
    C[i] = reshape(Ccat[pos+1 : pos+sub*h], sub, h)
    pos += sub * h
  end
  return C
end

"""
    train_pq(X, m, h, niter=25, V=false) -> C, B, error     (src/PQ.jl:68-99)
k-means per subspace on the device, kmeans++ seeding (`init=:kmpp`, src/PQ.jl:86) from the library's seeded stream.
"""
function train_pq(X::Matrix{Float32}, m::Integer, h::Integer, niter::Integer=25, V::Bool=false; seed::Integer=0)
  d, n = size(X)
  Ccat = Vector{Float32}(undef, h * d)
  B    = _result(Int16, m, n)
  err  = Ref{Cdouble}(0.0)
  _check(ccall((:rq_train_pq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Int16}, Ref{Cdouble}, Ptr{Cfloat}, Int64, Cint, Cint, Cint, Cint, UInt64),
    Ccat, B, err, X, Int64(n), Cint(d), Cint(m), Cint(h), Cint(niter), UInt64(seed)))
  return _split_codebooks(Ccat, d, Int(m), Int(h)), B, Float32(err[])
end

"""
    train_opq(X, m, h, niter, init, V=false) -> C, B, R, obj     (src/OPQ.jl:49-139)
"""
function train_opq(X::Matrix{Float32}, m::Integer, h::Integer, niter::Integer, init::String, V::Bool=false; seed::Integer=0)
  d, n = size(X)
  init in ("natural", "random") || error("Intialization $init unknown")   # src/OPQ.jl:74
  Ccat = Vector{Float32}(undef, h * d)
  B    = _result(Int16, m, n)
  R    = Matrix{Float32}(undef, d, d)
  obj  = zeros(Float32, niter + 1)
  _check(ccall((:rq_train_opq, librayuela_hip), Cint,
    (Ptr{Cfloat}, Ptr{Int16}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint, Cint, Cint, Cint, UInt64,
     Ptr{Cfloat}, Ptr{Cfloat}),
    Ccat, B, R, obj, X, Int64(n), Cint(d), Cint(m), Cint(h), Cint(niter), Cint(init == "natural" ? 0 : 1),
    UInt64(seed), C_NULL, C_NULL))
  return _split_codebooks(Ccat, d, Int(m), Int(h)), B, R, obj
end

# ---- resident handles (no counterpart in the reference: they remove its per-call marshalling) ----------------

"""
    HipIndex(C, d; devices=nothing)
Codes uploaded once, searched many times.  `devices = [0, 1, ...]`: one row shard per entry (RCCL gather of the
per-shard top-k + merge inside the library); `nothing`: the current device.  `search` returns what `linscan_pq`
returns (k-by-nq `dists`, ONE-based `idx`).
"""
mutable struct HipIndex
  h::Ptr{Cvoid}
  m::Int
  d::Int
  function HipIndex(C::Vector{Matrix{Cfloat}}, d::Int; devices::Union{Nothing,Vector{<:Integer}}=nothing)
    m = length(C)
    cen = cat(C..., dims=3)
    h = devices === nothing ?
      ccall((:rq_index_create, librayuela_hip), Ptr{Cvoid}, (Cint, Cint, Ptr{Cfloat}), Cint(m), Cint(d), cen) :
      ccall((:rq_index_create_sharded, librayuela_hip), Ptr{Cvoid}, (Cint, Cint, Ptr{Cfloat}, Ptr{Cint}, Cint),
            Cint(m), Cint(d), cen, convert(Vector{Cint}, devices), Cint(length(devices)))
    h == C_NULL && error("rq_index_create: " * unsafe_string(ccall((:rq_last_error, librayuela_hip), Cstring, ())))
    ix = new(h, m, d)
    finalizer(x -> (x.h != C_NULL && ccall((:rq_index_destroy, librayuela_hip), Cvoid, (Ptr{Cvoid},), x.h); x.h = C_NULL), ix)
    return ix
  end
end

function set_codes!(ix::HipIndex, B::Matrix{UInt8}; id_offset::Integer=0)      # B zero-based m-by-n
  _check(ccall((:rq_index_set_codes, librayuela_hip), Cint, (Ptr{Cvoid}, Ptr{Cuchar}, Int64, UInt32),
               ix.h, B, Int64(size(B, 2)), UInt32(id_offset)))
  return ix
end
set_codes!(ix::HipIndex, B::Matrix{T}; id_offset::Integer=0) where T <: Integer =
  set_codes!(ix, convert(Matrix{UInt8}, B .- 1); id_offset=id_offset)

function set_codes_synth!(ix::HipIndex, n::Integer, seed::Integer; id_offset::Integer=0)
  _check(ccall((:rq_index_set_codes_synth, librayuela_hip), Cint, (Ptr{Cvoid}, Int64, UInt64, UInt32),
               ix.h, Int64(n), UInt64(seed), UInt32(id_offset)))
  return ix
end

function search(ix::HipIndex, X::Matrix{Cfloat}, k::Int=10000; R::Union{Nothing,Matrix{Cfloat}}=nothing)
  d, nq = size(X)
  dists = Matrix{Cfloat}(undef, k, nq)
  res   = Matrix{Cuint}(undef, k, nq)
  if R === nothing
    _check(ccall((:rq_index_search, librayuela_hip), Cint,
      (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cfloat}, Int64, Cint, Cint), ix.h, dists, res, X, Int64(nq), Cint(k), Cint(1)))
  else
    _check(ccall((:rq_index_search_opq, librayuela_hip), Cint,
      (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cuint}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cint),
      ix.h, dists, res, X, R, Int64(nq), Cint(k), Cint(1)))
  end
  return dists, res
end

"""
    HipDataset(X)
`X` (d-by-n) uploaded once; `quantize(ds, C)` == `quantize_pq(X, C)`, `quantize(ds, C; R=R)` == `quantize_opq(X, R, C)`.
"""
mutable struct HipDataset
  h::Ptr{Cvoid}
  n::Int
  function HipDataset(X::Matrix{Float32})
    d, n = size(X)
    h = ccall((:rq_dataset_upload, librayuela_hip), Ptr{Cvoid}, (Ptr{Cfloat}, Int64, Cint), X, Int64(n), Cint(d))
    h == C_NULL && error("rq_dataset_upload: " * unsafe_string(ccall((:rq_last_error, librayuela_hip), Cstring, ())))
    ds = new(h, n)
    finalizer(x -> (x.h != C_NULL && ccall((:rq_dataset_free, librayuela_hip), Cvoid, (Ptr{Cvoid},), x.h); x.h = C_NULL), ds)
    return ds
  end
end

function quantize(ds::HipDataset, C::Vector{Matrix{Float32}}; R::Union{Nothing,Matrix{Float32}}=nothing)
  m, h = length(C), size(C[1], 2)
  B = Matrix{Int16}(undef, m, ds.n)
  _check(ccall((:rq_dataset_encode, librayuela_hip), Cint,
    (Ptr{Cvoid}, Ptr{Cuchar}, Ptr{Int16}, Ptr{Cfloat}, Ptr{Cfloat}, Cint, Cint),
    ds.h, C_NULL, B, R === nothing ? C_NULL : R, _cat_codebooks(C), Cint(m), Cint(h)))
  return B
end

end # module
