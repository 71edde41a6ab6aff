# gen_encode_golden.jl -- pins the encode half of the path against the REAL reference stack.
#
#   julia julia/gen_encode_golden.jl [tests/golden/bin]
#
# Needs Julia 1.x with Distances v0.8.0 and Clustering v0.12.2 (the versions Rayuela.jl's Manifest.toml:79-83,
# 135-139 pins; `]add Distances@0.8.0 Clustering@0.12.2`).  If Rayuela.jl itself is installed its own
# quantize_pq / quantize_opq / quantize_rvq are called; otherwise the two package calls those functions make
# (src/PQ.jl:40-41, src/OPQ.jl:26, src/RVQ.jl:40-56) are issued directly on the same sub-matrices.
#
# Reads the raw mirrors of the golden inputs written by tests/export_golden_bin.py (cases.txt, <case>.X.f32,
# <case>.C.f32, <case>.R.f32) and writes, next to them,
#   <case>.codes_julia.i16       m x n Int16, ONE-based  (quantize_pq / quantize_rvq)
#   <case>.codes_opq_julia.i16   m x n Int16, ONE-based  (quantize_opq, cases with R)
# tests/test_julia_golden.py compares them with the oracle's codes and classifies every difference by the stored
# float64 top-2 gap (a difference with a gap above the f32 error bound is a real parity failure).
#
# NOTE: the build image has no Julia, so this script has not been executed there.
using LinearAlgebra
import Distances, Clustering

const HAVE_RAYUELA = try
  @eval import Rayuela
  true
catch
  false
end

# src/utils.jl:179-203: contiguous ranges, the first d % m ranges get one extra element
function split_ranges(d::Int, m::Int)
  per, extra = divrem(d, m)
  out = UnitRange{Int}[]
  lo = 1
  for i in 1:m
    len = per + (i <= extra ? 1 : 0)
    push!(out, lo:(lo + len - 1))
    lo += len
  end
  out
end

function assign!(codes::Vector{Int}, Ci::Matrix{Float32}, Xs::Matrix{Float32})
  n, h = size(Xs, 2), size(Ci, 2)
  dmat = Distances.pairwise(Distances.SqEuclidean(), Ci, Xs)                                   # src/PQ.jl:40
  Clustering.update_assignments!(dmat, true, codes, zeros(Float32, n), zeros(Int, h), zeros(Bool, h), Int[])  # :41
  codes
end

function encode_pq(X::Matrix{Float32}, C::Vector{Matrix{Float32}})
  HAVE_RAYUELA && return Rayuela.quantize_pq(X, C)
  d, n = size(X); m = length(C)
  B = Matrix{Int16}(undef, m, n)
  for (i, r) in enumerate(split_ranges(d, m))
    B[i, :] = assign!(zeros(Int, n), C[i], X[r, :])
  end
  B
end

encode_opq(X, R, C) = HAVE_RAYUELA ? Rayuela.quantize_opq(X, R, C) : encode_pq(R' * X, C)    # src/OPQ.jl:26

function encode_rvq(X::Matrix{Float32}, C::Vector{Matrix{Float32}})
  HAVE_RAYUELA && return Rayuela.quantize_rvq(X, C)[1]
  d, n = size(X); m = length(C)
  B = Matrix{Int16}(undef, m, n)
  Xr = copy(X)
  for i in 1:m
    codes = assign!(zeros(Int, n), C[i], Xr)                                                   # src/RVQ.jl:40-47
    B[i, :] = codes
    Xr .-= C[i][:, codes]                                                                      # :56
  end
  B
end

readf32(path, dims...) = reshape(reinterpret(Float32, read(path)), dims...) |> collect

function main(dir)
  for line in eachline(joinpath(dir, "cases.txt"))
    isempty(strip(line)) && continue
    name, kind, n, d, m, h, hasR = split(line)
    n, d, m, h = parse.(Int, (n, d, m, h))
    X = readf32(joinpath(dir, "$name.X.f32"), d, n)
    craw = reinterpret(Float32, read(joinpath(dir, "$name.C.f32")))
    C = Matrix{Float32}[]
    pos = 0
    widths = kind == "pq" ? length.(split_ranges(d, m)) : fill(d, m)
    for w in widths
      push!(C, collect(reshape(craw[pos+1:pos+w*h], w, h)))
      pos += w * h
    end
    B = kind == "pq" ? encode_pq(X, C) : encode_rvq(X, C)
    write(joinpath(dir, "$name.codes_julia.i16"), convert(Matrix{Int16}, B))
    println("$name: $(size(B)) codes written")
    if hasR == "1"
      R = readf32(joinpath(dir, "$name.R.f32"), d, d)
      Bo = encode_opq(X, R, C)
      write(joinpath(dir, "$name.codes_opq_julia.i16"), convert(Matrix{Int16}, Bo))
      println("$name: OPQ codes written")
    end
  end
end

main(length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden", "bin"))
