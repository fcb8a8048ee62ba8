# WaveletsExtHIP.jl -- reference-side binding for libwaveletsext_hip.so.
#
# Delivered as source: there is no Julia in the build image or on the GPU box, so this file is NOT executed by the test
# suite.  What is checked without Julia (tests/test_julia_shim_static.py): the `ccall` tuples live in libwx.jl, which
# is generated from include/waveletsext_hip.h and compared with it; every call site `wx_name(...)` below is parsed and
# its argument count compared with the header's prototype; every entry family of the header must have a caller here.
# Every entry point itself is exercised through the same C ABI by tests/ via ctypes.
#
# The module adds methods to the generic functions of Wavelets.jl / WaveletsExt.jl for a marker wrapper type `HIP(x)`,
# so existing user code switches the hot path by wrapping an argument:
#
#     using Wavelets, WaveletsExt, WaveletsExtHIP
#     xw = wpdall(HIP(x), wt, L)            # MI355X kernels instead of dwt/dwt_all.jl:260-282
#     x̂  = iwpdall(HIP(xw), wt, tree)
#     wpt!(HIP(y), x, wt, tree)             # `!` forms dispatch on the wrapped OUTPUT and return it
#     tr = bestbasistree(HIP(x), wt, 11, JBB(redundant = true))   # acwpd + JBB without the (n, 4095, N) table
#
# Allocating forms take the wrapper on their (first) data argument and return a plain array of the parent's kind;
# mutating forms take it on the array they fill and return that argument, like the reference's.
# Plain `Array`s cross the boundary as host pointers (the library stages H2D / D2H); AMDGPU.jl `ROCArray`s cross as
# device pointers and the call is asynchronous on `WaveletsExtHIP.stream()`.
# Errors: the library reports the reference's assertion / argument failures as status codes, `check` rethrows them as
# AssertionError / ArgumentError / BoundsError, so `@test_throws` expectations (test/transforms.jl:63-65) keep holding.
module WaveletsExtHIP

using Wavelets, WaveletsExt
using Statistics: mean, median
import Wavelets.Transforms: wpt, wpt!, iwpt, iwpt!, dwt, idwt
import Wavelets.Threshold: bestbasistree, noisest, HardTH, SoftTH, SemiSoftTH, SteinTH, THType
import WaveletsExt.DWT: wpd, wpd!, iwpd, iwpd!, wpdall, iwpdall, wptall, iwptall, dwtall, idwtall
import WaveletsExt.SWT: sdwt, sdwt!, isdwt, isdwt!, swpt, swpt!, iswpt, iswpt!, swpd, swpd!, iswpd, iswpd!,
                        sdwtall, isdwtall, swptall, iswptall, swpdall, iswpdall
import WaveletsExt.ACWT: acdwt, acdwt!, iacdwt, iacdwt!, acwpt, acwpt!, iacwpt, iacwpt!, acwpd, acwpd!, iacwpd, iacwpd!,
                         acdwtall, iacdwtall, acwptall, iacwptall, acwpdall, iacwpdall
import WaveletsExt.BestBasis: tree_costs, JBB, BB, LoglpCost, NormCost, ShannonEntropyCost, LogEnergyEntropyCost,
                              bestbasis_treeselection, bestbasistreeall
import WaveletsExt.Utils: getbasiscoef, getbasiscoefall
import WaveletsExt.Denoising: surethreshold, relerrorthreshold
import WaveletsExt.LDB: energy_map, discriminant_power, TimeFrequency, ProbabilityDensity, Signatures,
                        FishersClassSeparability, RobustFishersClassSeparability

export HIP

include("libwx.jl")          # raw bindings, generated from include/waveletsext_hip.h

# ---------------------------------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------------------------------
"Marker wrapper: dispatches the hot path to the HIP library.  `HIP(a)` shares `a`'s memory."
struct HIP{T,N,A<:AbstractArray{T,N}} <: AbstractArray{T,N}
    a::A
end
Base.size(x::HIP) = size(x.a)
Base.getindex(x::HIP, i...) = getindex(x.a, i...)
Base.setindex!(x::HIP, v, i...) = setindex!(x.a, v, i...)
Base.parent(x::HIP) = x.a
raw(x::HIP) = x.a
raw(x) = x

const FT = Union{Float64,Float32}
const WX_EASSERT, WX_EARG, WX_EBOUNDS = Cint(-1), Cint(-2), Cint(-3)
const STREAM = Ref{Ptr{Cvoid}}(C_NULL)
"hipStream_t the device-pointer calls are queued on (C_NULL = default stream)"
stream() = STREAM[]
stream!(s) = (STREAM[] = convert(Ptr{Cvoid}, s))

function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(wx_last_error())
    rc == WX_EASSERT && throw(AssertionError(msg))
    rc == WX_EARG && throw(ArgumentError(msg))
    rc == WX_EBOUNDS && throw(BoundsError())
    error("libwaveletsext_hip status $rc: $msg")
end

version() = Int(wx_version())
device_count() = Int(wx_device_count())
build_info() = unsafe_string(wx_build_info())
shutdown() = check(wx_shutdown())
"MADV_HUGEPAGE advice on host result arrays (on by default, persists on the address range); returns the previous setting"
set_host_hugepages(on::Bool) = wx_set_host_hugepages(Cint(on)) != 0

qmfvec(wt::OrthoFilter) = Vector{Float64}(WT.qmf(wt))
"array of the parent's kind (Array stays Array, ROCArray stays ROCArray)"
newlike(x, ::Type{T}, dims) where T = similar(raw(x), T, dims)
"`L | tree` argument -> (L, tree bytes or C_NULL, number of tree bytes); a BitVector crosses as one byte per node"
treearg(L::Integer) = (Int(L), C_NULL, 0)
function treearg(tree::BitVector)
    tb = Vector{UInt8}(tree)
    return (0, tb, length(tb))
end
smarg(::Nothing) = -1                 # average-based inverse
smarg(sm::Integer) = Int(sm)          # shift-based inverse with shift sm

# ---------------------------------------------------------------------------------------------------------------------
# one line per C entry family: `sig` is the size of one signal, its length selects the 1-D or the 2-D entry point
# ---------------------------------------------------------------------------------------------------------------------
c_wpd(T, x, y, sig::NTuple{1,Int}, L, N, q) = check(wx_wpd1d(T, x, y, sig[1], L, N, q, length(q), stream()))
c_wpd(T, x, y, sig::NTuple{2,Int}, L, N, q) = check(wx_wpd2d(T, x, y, sig[1], sig[2], L, N, q, length(q), stream()))
c_wpt(T, x, y, sig::NTuple{1,Int}, L, tb, nt, N, q) = check(wx_wpt1d(T, x, y, sig[1], L, tb, nt, N, q, length(q), stream()))
c_wpt(T, x, y, sig::NTuple{2,Int}, L, tb, nt, N, q) = check(wx_wpt2d(T, x, y, sig[1], sig[2], L, tb, nt, N, q, length(q), stream()))
c_iwpt(T, xw, x, sig::NTuple{1,Int}, L, tb, nt, N, q) = check(wx_iwpt1d(T, xw, x, sig[1], L, tb, nt, N, q, length(q), stream()))
c_iwpt(T, xw, x, sig::NTuple{2,Int}, L, tb, nt, N, q) = check(wx_iwpt2d(T, xw, x, sig[1], sig[2], L, tb, nt, N, q, length(q), stream()))
c_iwpd(T, xw, x, sig::NTuple{1,Int}, k, L, tb, nt, N, q) = check(wx_iwpd1d(T, xw, x, sig[1], k, L, tb, nt, N, q, length(q), stream()))
c_iwpd(T, xw, x, sig::NTuple{2,Int}, k, L, tb, nt, N, q) = check(wx_iwpd2d(T, xw, x, sig[1], sig[2], k, L, tb, nt, N, q, length(q), stream()))
c_getbasiscoef(T, Xw, out, sig::NTuple{1,Int}, k, tb, nt, N) = check(wx_getbasiscoef1d(T, Xw, out, sig[1], k, tb, nt, N, stream()))
c_getbasiscoef(T, Xw, out, sig::NTuple{2,Int}, k, tb, nt, N) = check(wx_getbasiscoef2d(T, Xw, out, sig[1], sig[2], k, tb, nt, N, stream()))
c_getbasiscoef_trees(T, Xw, out, sig::NTuple{1,Int}, k, tb, nt, N) = check(wx_getbasiscoef1d_trees(T, Xw, out, sig[1], k, tb, nt, N, stream()))
c_getbasiscoef_trees(T, Xw, out, sig::NTuple{2,Int}, k, tb, nt, N) =
    check(wx_getbasiscoef2d_trees(T, Xw, out, sig[1], sig[2], k, tb, nt, N, stream()))
c_dwt3d(T, x, y, sig::NTuple{3,Int}, L, N, q) = check(wx_dwt3d(T, x, y, sig[1], sig[2], sig[3], L, N, q, length(q), stream()))
c_idwt3d(T, x, y, sig::NTuple{3,Int}, L, N, q) = check(wx_idwt3d(T, x, y, sig[1], sig[2], sig[3], L, N, q, length(q), stream()))

c_sdwt(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_sdwt1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_sdwt(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_sdwt2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_swpt(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_swpt1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_swpt(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_swpt2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_swpd(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_swpd1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_swpd(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_swpd2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_isdwt(T, xw, x, sig::NTuple{1,Int}, L, sm, N, q) = check(wx_isdwt1d(T, xw, x, sig[1], L, sm, N, q, length(q), stream()))
c_isdwt(T, xw, x, sig::NTuple{2,Int}, L, sm, N, q) = check(wx_isdwt2d(T, xw, x, sig[1], sig[2], L, sm, N, q, length(q), stream()))
c_iswpt(T, xw, x, sig::NTuple{1,Int}, L, sm, N, q) = check(wx_iswpt1d(T, xw, x, sig[1], L, sm, N, q, length(q), stream()))
c_iswpt(T, xw, x, sig::NTuple{2,Int}, L, sm, N, q) = check(wx_iswpt2d(T, xw, x, sig[1], sig[2], L, sm, N, q, length(q), stream()))
c_iswpd(T, xw, x, sig::NTuple{1,Int}, k, L, tb, nt, sm, N, q) = check(wx_iswpd1d(T, xw, x, sig[1], k, L, tb, nt, sm, N, q, length(q), stream()))
c_iswpd(T, xw, x, sig::NTuple{2,Int}, k, L, tb, nt, sm, N, q) = check(wx_iswpd2d(T, xw, x, sig[1], sig[2], k, L, tb, nt, sm, N, q, length(q), stream()))

# autocorrelation family: Float64 only, like the reference (acwt/acwt_one_level.jl:101-106) -- a Float32 call is a
# MethodError here as it is there
c_acdwt(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_acdwt1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_acdwt(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_acdwt2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_acwpt(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_acwpt1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_acwpt(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_acwpt2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_acwpd(T, x, xw, sig::NTuple{1,Int}, L, N, q) = check(wx_acwpd1d(T, x, xw, sig[1], L, N, q, length(q), stream()))
c_acwpd(T, x, xw, sig::NTuple{2,Int}, L, N, q) = check(wx_acwpd2d(T, x, xw, sig[1], sig[2], L, N, q, length(q), stream()))
c_iacdwt(T, xw, x, sig::NTuple{1,Int}, L, N) = check(wx_iacdwt1d(T, xw, x, sig[1], L, N, stream()))
c_iacdwt(T, xw, x, sig::NTuple{2,Int}, L, N) = check(wx_iacdwt2d(T, xw, x, sig[1], sig[2], L, N, stream()))
c_iacwpt(T, xw, x, sig::NTuple{1,Int}, L, N) = check(wx_iacwpt1d(T, xw, x, sig[1], L, N, stream()))
c_iacwpt(T, xw, x, sig::NTuple{2,Int}, L, N) = check(wx_iacwpt2d(T, xw, x, sig[1], sig[2], L, N, stream()))
c_iacwpd(T, xw, x, sig::NTuple{1,Int}, k, L, tb, nt, N) = check(wx_iacwpd1d(T, xw, x, sig[1], k, L, tb, nt, N, stream()))
c_iacwpd(T, xw, x, sig::NTuple{2,Int}, k, L, tb, nt, N) = check(wx_iacwpd2d(T, xw, x, sig[1], sig[2], k, L, tb, nt, N, stream()))

c_jbb_costs(T, s, q2, Ntot, sig::NTuple{1,Int}, k, red, kind, p, costs) = check(wx_jbb_costs(T, s, q2, Ntot, sig[1], k, red, kind, p, costs, stream()))
c_jbb_costs(T, s, q2, Ntot, sig::NTuple{2,Int}, k, red, kind, p, costs) = check(wx_jbb_costs2d(T, s, q2, Ntot, sig[1], sig[2], k, red, kind, p, costs, stream()))
c_bb_costs(T, X, costs, sig::NTuple{1,Int}, k, N, red, kind) = check(wx_bb_costs(T, X, costs, sig[1], k, N, red, kind, stream()))
c_bb_costs(T, X, costs, sig::NTuple{2,Int}, k, N, red, kind) = check(wx_bb_costs2d(T, X, costs, sig[1], sig[2], k, N, red, kind, stream()))
c_treeselect(T, costs, k, sig::NTuple{1,Int}, tmax, tree) = check(wx_treeselect(T, costs, k, sig[1], tmax, tree))
c_treeselect(T, costs, k, sig::NTuple{2,Int}, tmax, tree) = check(wx_treeselect2d(T, costs, k, sig[1], sig[2], tmax, tree))

maxL(sig) = maxtransformlevels(minimum(sig))
treelen(sig::NTuple{1,Int}) = sig[1] - 1
treelen(sig::NTuple{2,Int}) = gettreelength(sig...)

# ---------------------------------------------------------------------------------------------------------------------
# Method signatures.  Every method below has the signature of the reference method it stands in for with `HIP` put on
# the dispatched array and NOTHING widened: same dimensionality per method (`HIP{T,1}` where the reference says
# `AbstractVector{T}`), `L::Integer` and `tree::BitVector` as separate methods, `sm::Integer` present or absent.  A
# signature that is wider than the reference's in one argument and narrower in another would be ambiguous with it.
# ---------------------------------------------------------------------------------------------------------------------

# ---------------------------------------------------------------------------------------------------------------------
# decimated packets -- DWT.jl, dwt/dwt_all.jl (1-D wpt / iwpt: Wavelets.jl)
# ---------------------------------------------------------------------------------------------------------------------
for N in 1:2
    @eval begin
        # wpd! (DWT.jl:131-209): y (sz..., L+1) from x (sz...)
        function wpd!(y::HIP{T,$(N + 1)}, x::AbstractArray{T,$N}, wt::OrthoFilter, L::Integer = maxtransformlevels(x)) where T<:FT
            @assert size(y) == (size(x)..., L + 1)
            c_wpd(T, raw(x), raw(y), size(x), L, 1, qmfvec(wt))
            return y
        end
        # wpd (DWT.jl:60-88)
        function wpd(x::HIP{T,$N}, wt::OrthoFilter, L::Integer = maxtransformlevels(x)) where T<:FT
            y = newlike(x, T, (size(x)..., L + 1))
            c_wpd(T, raw(x), y, size(x), L, 1, qmfvec(wt))
            return y
        end
        # iwpd! by level (DWT.jl:322-337): the level defaults to the deepest one
        iwpd!(x̂::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::OrthoFilter) where T<:FT = iwpd!(x̂, xw, wt, maxtransformlevels(x̂))
        # wpt! / iwpt! by level: defaults (2-D: DWT.jl:493-498, 655-660)
        wpt!(y::HIP{T,$N}, x::AbstractArray{T,$N}, wt::OrthoFilter) where T<:FT = wpt!(y, x, wt, maxtransformlevels(x))
        iwpt!(x̂::HIP{T,$N}, xw::AbstractArray{T,$N}, wt::OrthoFilter) where T<:FT = iwpt!(x̂, xw, wt, maxtransformlevels(xw))
        wpt(x::HIP{T,$N}, wt::OrthoFilter) where T<:FT = wpt(x, wt, maxtransformlevels(x))
        iwpt(xw::HIP{T,$N}, wt::OrthoFilter) where T<:FT = iwpt(xw, wt, maxtransformlevels(xw))
    end
    for AT in (:Integer, :BitVector)
        @eval begin
            # iwpd! by level or by tree (DWT.jl:322-401): x̂ (sz...) from xw (sz..., k)
            function iwpd!(x̂::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::OrthoFilter, arg::$AT) where T<:FT
                @assert size(x̂) == size(xw)[1:end-1]
                L, tb, nt = treearg(arg)
                c_iwpd(T, raw(xw), raw(x̂), size(x̂), size(xw)[end], L, tb, nt, 1, qmfvec(wt))
                return x̂
            end
            # wpt! / iwpt! (2-D: DWT.jl:493-548, 655-710; 1-D: the Wavelets.jl methods wptall / iwptall call, dwt_all.jl:162,221)
            function wpt!(y::HIP{T,$N}, x::AbstractArray{T,$N}, wt::OrthoFilter, arg::$AT) where T<:FT
                @assert size(y) == size(x)
                L, tb, nt = treearg(arg)
                c_wpt(T, raw(x), raw(y), size(x), L, tb, nt, 1, qmfvec(wt))
                return y
            end
            function iwpt!(x̂::HIP{T,$N}, xw::AbstractArray{T,$N}, wt::OrthoFilter, arg::$AT) where T<:FT
                @assert size(x̂) == size(xw)
                L, tb, nt = treearg(arg)
                c_iwpt(T, raw(xw), raw(x̂), size(xw), L, tb, nt, 1, qmfvec(wt))
                return x̂
            end
            # wpt / iwpt (DWT.jl:440-451, 594-605)
            function wpt(x::HIP{T,$N}, wt::OrthoFilter, arg::$AT) where T<:FT
                y = newlike(x, T, size(x))
                wpt!(HIP(y), x, wt, arg)
                return y
            end
            function iwpt(xw::HIP{T,$N}, wt::OrthoFilter, arg::$AT) where T<:FT
                x̂ = newlike(xw, T, size(xw))
                iwpt!(HIP(x̂), xw, wt, arg)
                return x̂
            end
        end
    end
end
# wpdall (dwt/dwt_all.jl:260-282)
function wpdall(x::HIP{T}, wt::OrthoFilter, L::Integer = maxL(size(x)[1:end-1])) where T<:FT
    @assert 2 ≤ ndims(x) ≤ 3
    sz = size(x)[1:end-1]; N = size(x)[end]
    y = newlike(x, T, (sz..., L + 1, N))
    c_wpd(T, raw(x), y, sz, L, N, qmfvec(wt))
    return y
end
# by level, the level defaulting to the deepest one (DWT.jl:257-265, dwt_all.jl:152-166, 210-225, 324-342)
iwpd(xw::HIP{T}, wt::OrthoFilter) where T<:FT = iwpd(xw, wt, maxL(size(xw)[1:end-1]))
iwpdall(xw::HIP{T}, wt::OrthoFilter) where T<:FT = iwpdall(xw, wt, maxL(size(xw)[1:end-2]))
wptall(x::HIP{T}, wt::OrthoFilter) where T<:FT = wptall(x, wt, maxL(size(x)[1:end-1]))
iwptall(xw::HIP{T}, wt::OrthoFilter) where T<:FT = iwptall(xw, wt, maxL(size(xw)[1:end-1]))
for AT in (:Integer, :BitVector)
    @eval begin
        # iwpd (DWT.jl:257-274)
        function iwpd(xw::HIP{T}, wt::OrthoFilter, arg::$AT) where T<:FT
            @assert 2 ≤ ndims(xw) ≤ 3
            sz = size(xw)[1:end-1]
            x̂ = newlike(xw, T, sz)
            L, tb, nt = treearg(arg)
            c_iwpd(T, raw(xw), x̂, sz, size(xw)[end], L, tb, nt, 1, qmfvec(wt))
            return x̂
        end
        # iwpdall (dwt/dwt_all.jl:324-342)
        function iwpdall(xw::HIP{T}, wt::OrthoFilter, arg::$AT) where T<:FT
            @assert 3 ≤ ndims(xw) ≤ 4
            sz = size(xw)[1:end-2]; k = size(xw)[end-1]; N = size(xw)[end]
            x̂ = newlike(xw, T, (sz..., N))
            L, tb, nt = treearg(arg)
            c_iwpd(T, raw(xw), x̂, sz, k, L, tb, nt, N, qmfvec(wt))
            return x̂
        end
        # wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225)
        function wptall(x::HIP{T}, wt::OrthoFilter, arg::$AT) where T<:FT
            @assert 2 ≤ ndims(x) ≤ 3
            sz = size(x)[1:end-1]; N = size(x)[end]
            y = newlike(x, T, size(x))
            L, tb, nt = treearg(arg)
            c_wpt(T, raw(x), y, sz, L, tb, nt, N, qmfvec(wt))
            return y
        end
        function iwptall(xw::HIP{T}, wt::OrthoFilter, arg::$AT) where T<:FT
            @assert 2 ≤ ndims(xw) ≤ 3
            sz = size(xw)[1:end-1]; N = size(xw)[end]
            x̂ = newlike(xw, T, size(xw))
            L, tb, nt = treearg(arg)
            c_iwpt(T, raw(xw), x̂, sz, L, tb, nt, N, qmfvec(wt))
            return x̂
        end
    end
end

# dwt / idwt / dwtall / idwtall (dwt/dwt_all.jl:39-54, 95-110): the pyramid is the packet transform along the :dwt
# tree (test/transforms.jl:42), which the library runs on its own pyramid kernels; cubes (4-D batches) take wx_dwt3d.
dwttree(sig, L) = maketree(sig..., L, :dwt)
function dwt(x::HIP{T}, wt::OrthoFilter, L::Integer = maxtransformlevels(x)) where T<:FT
    ndims(x) == 3 && return dropdims(dwtall(HIP(reshape(raw(x), size(x)..., 1)), wt, L), dims = 4)
    return L == 0 ? wpt(x, wt, 0) : wpt(x, wt, dwttree(size(x), L))
end
function idwt(xw::HIP{T}, wt::OrthoFilter, L::Integer = maxtransformlevels(xw)) where T<:FT
    ndims(xw) == 3 && return dropdims(idwtall(HIP(reshape(raw(xw), size(xw)..., 1)), wt, L), dims = 4)
    return L == 0 ? iwpt(xw, wt, 0) : iwpt(xw, wt, dwttree(size(xw), L))
end
function dwtall(x::HIP{T}, wt::OrthoFilter, L::Integer = maxL(size(x)[1:end-1])) where T<:FT
    @assert 2 ≤ ndims(x) ≤ 4
    sz = size(x)[1:end-1]
    if length(sz) == 3
        y = newlike(x, T, size(x))
        c_dwt3d(T, raw(x), y, sz, L, size(x)[end], qmfvec(wt))
        return y
    end
    return L == 0 ? wptall(x, wt, 0) : wptall(x, wt, dwttree(sz, L))
end
function idwtall(xw::HIP{T}, wt::OrthoFilter, L::Integer = maxL(size(xw)[1:end-1])) where T<:FT
    @assert 2 ≤ ndims(xw) ≤ 4
    sz = size(xw)[1:end-1]
    if length(sz) == 3
        y = newlike(xw, T, size(xw))
        c_idwt3d(T, raw(xw), y, sz, L, size(xw)[end], qmfvec(wt))
        return y
    end
    return L == 0 ? iwptall(xw, wt, 0) : iwptall(xw, wt, dwttree(sz, L))
end

# getbasiscoef / getbasiscoefall (Utils.jl:101-225): gather of the leaves out of a packet table, on the device
function getbasiscoef(Xw::HIP{T}, tree::BitVector) where T<:FT
    @assert 2 ≤ ndims(Xw) ≤ 3
    sz = size(Xw)[1:end-1]
    out = newlike(Xw, T, sz)
    _, tb, nt = treearg(tree)
    c_getbasiscoef(T, raw(Xw), out, sz, size(Xw)[end], tb, nt, 1)
    return out
end
function getbasiscoefall(Xw::HIP{T}, tree::BitVector) where T<:FT
    @assert 3 ≤ ndims(Xw) ≤ 4
    sz = size(Xw)[1:end-2]; k = size(Xw)[end-1]; N = size(Xw)[end]
    out = newlike(Xw, T, (sz..., N))
    _, tb, nt = treearg(tree)
    c_getbasiscoef(T, raw(Xw), out, sz, k, tb, nt, N)
    return out
end
# one tree per signal (Utils.jl:199-225), the consumer of bestbasistreeall: the BitMatrix goes over as bytes and ONE launch
# gathers every signal (wx_getbasiscoef*_trees checks each tree like the reference's `@assert all(mapslices(isvalidtree, ...))`)
function getbasiscoefall(Xw::HIP{T}, trees::BitMatrix) where T<:FT
    @assert 3 ≤ ndims(Xw) ≤ 4
    sz = size(Xw)[1:end-2]; k = size(Xw)[end-1]; N = size(Xw)[end]
    @assert size(trees, 2) == N
    out = newlike(Xw, T, (sz..., N))
    tb = Matrix{UInt8}(trees)
    c_getbasiscoef_trees(T, raw(Xw), out, sz, k, tb, size(tb, 1), N)
    return out
end

# ---------------------------------------------------------------------------------------------------------------------
# stationary transforms -- SWT.jl, swt/swt_all.jl (1-D and 2-D)
# ---------------------------------------------------------------------------------------------------------------------
sdwt_cols(sig, L) = length(sig) == 1 ? L + 1 : 3 * L + 1                    # SWT.jl:68
swpt_cols(sig, L) = length(sig) == 1 ? (1 << L) : (1 << (2 * L))            # SWT.jl:401, swt_all.jl:163
swpd_cols(sig, L) = length(sig) == 1 ? (1 << (L + 1)) - 1 : ((1 << (2 * (L + 1))) - 1) ÷ 3   # SWT.jl:800
sdwt_levels(sig, k) = length(sig) == 1 ? k - 1 : (k - 1) ÷ 3                # SWT.jl:265, 292
function swpt_levels(sig, k)                                                # SWT.jl:619, 653
    if length(sig) == 1
        isdyadic(k) || throw(ArgumentError("Number of columns of xw is not dyadic."))
        return ndyadicscales(k)
    end
    L = ndyadicscales(k) ÷ 2
    (1 << (2 * L)) == k || throw(ArgumentError("Size of dimension 3 is not a power of 4."))
    return L
end

# forward transforms, stationary and autocorrelation (the containers are the same): sdwt!/swpt!/swpd! (SWT.jl:109-158,
# 439-513, 840-902), acdwt!/acwpt!/acwpd! (ACWT.jl:109-157, 427-501, 733-793), the allocating forms (SWT.jl:60-74,
# 390-406, 790-806; ACWT.jl:60-75, 379-394, 683-698) and the batch drivers (swt_all.jl:33-50, 156-176, 279-299;
# acwt_all.jl:33-50, 136-153, 239-259).  The autocorrelation entries exist for Float64 only, like the reference's
# (acwt/acwt_one_level.jl:101-106): a Float32 call is a MethodError here as it is there.
for (f, f!, fall, cols, cfun) in ((:sdwt, :sdwt!, :sdwtall, :sdwt_cols, :c_sdwt),
                                  (:swpt, :swpt!, :swptall, :swpt_cols, :c_swpt),
                                  (:swpd, :swpd!, :swpdall, :swpd_cols, :c_swpd),
                                  (:acdwt, :acdwt!, :acdwtall, :sdwt_cols, :c_acdwt),
                                  (:acwpt, :acwpt!, :acwptall, :swpt_cols, :c_acwpt),
                                  (:acwpd, :acwpd!, :acwpdall, :swpd_cols, :c_acwpd))
    for N in 1:2
        @eval function $f!(xw::HIP{T,$(N + 1)}, x::AbstractArray{T,$N}, wt::OrthoFilter, L::Integer = maxtransformlevels(x)) where T<:FT
            @assert size(xw) == (size(x)..., $cols(size(x), L))
            $cfun(T, raw(x), raw(xw), size(x), L, 1, qmfvec(wt))
            return xw
        end
    end
    @eval begin
        function $f(x::HIP{T}, wt::OrthoFilter, L::Integer = maxtransformlevels(x)) where T<:FT
            @assert 1 ≤ ndims(x) ≤ 2
            xw = newlike(x, T, (size(x)..., $cols(size(x), L)))
            $cfun(T, raw(x), xw, size(x), L, 1, qmfvec(wt))
            return xw
        end
        function $fall(x::HIP{T}, wt::OrthoFilter, L::Integer = maxL(size(x)[1:end-1])) where T<:FT
            @assert 2 ≤ ndims(x) ≤ 3
            sz = size(x)[1:end-1]; N = size(x)[end]
            xw = newlike(x, T, (sz..., $cols(sz, L), N))
            $cfun(T, raw(x), xw, sz, L, N, qmfvec(wt))
            return xw
        end
    end
end

# isdwt / iswpt: average-based without `sm`, shift-based with it (SWT.jl:197-358, 551-758; swt_all.jl:89-122, 212-245)
for (f, f!, fall, levels, cfun) in ((:isdwt, :isdwt!, :isdwtall, :sdwt_levels, :c_isdwt),
                                    (:iswpt, :iswpt!, :iswptall, :swpt_levels, :c_iswpt))
    for smsig in ((), (:(sm::Integer),))              # without / with the shift argument
        smval = isempty(smsig) ? :(nothing) : :sm
        for N in 1:2
            @eval function $f!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::OrthoFilter, $(smsig...)) where T<:FT
                @assert size(x) == size(xw)[1:end-1]
                $cfun(T, raw(xw), raw(x), size(x), $levels(size(x), size(xw)[end]), smarg($smval), 1, qmfvec(wt))
                return x
            end
        end
        @eval begin
            function $f(xw::HIP{T}, wt::OrthoFilter, $(smsig...)) where T<:FT
                @assert 2 ≤ ndims(xw) ≤ 3
                sz = size(xw)[1:end-1]
                x = newlike(xw, T, sz)
                $cfun(T, raw(xw), x, sz, $levels(sz, size(xw)[end]), smarg($smval), 1, qmfvec(wt))
                return x
            end
            function $fall(xw::HIP{T}, wt::OrthoFilter, $(smsig...)) where T<:FT
                @assert 3 ≤ ndims(xw) ≤ 4
                sz = size(xw)[1:end-2]; N = size(xw)[end]
                x = newlike(xw, T, (sz..., N))
                $cfun(T, raw(xw), x, sz, $levels(sz, size(xw)[end-1]), smarg($smval), N, qmfvec(wt))
                return x
            end
        end
    end
end

# iswpd by level or tree, average- or shift-based (SWT.jl:952-1199, swt_all.jl:343-392)
iswpd(xw::HIP{T}, wt::OrthoFilter) where T<:FT = iswpd(xw, wt, maxL(size(xw)[1:end-1]))
iswpdall(xw::HIP{T}, wt::OrthoFilter) where T<:FT = iswpdall(xw, wt, maxL(size(xw)[1:end-2]))
for N in 1:2
    @eval iswpd!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::OrthoFilter) where T<:FT = iswpd!(x, xw, wt, maxL(size(x)))
end
for AT in (:Integer, :BitVector), smsig in ((), (:(sm::Integer),))
    smval = isempty(smsig) ? :(nothing) : :sm
    for N in 1:2
        @eval function iswpd!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::OrthoFilter, arg::$AT, $(smsig...)) where T<:FT
            @assert size(x) == size(xw)[1:end-1]
            L, tb, nt = treearg(arg)
            c_iswpd(T, raw(xw), raw(x), size(x), size(xw)[end], L, tb, nt, smarg($smval), 1, qmfvec(wt))
            return x
        end
    end
    @eval begin
        function iswpd(xw::HIP{T}, wt::OrthoFilter, arg::$AT, $(smsig...)) where T<:FT
            @assert 2 ≤ ndims(xw) ≤ 3
            sz = size(xw)[1:end-1]
            x = newlike(xw, T, sz)
            L, tb, nt = treearg(arg)
            c_iswpd(T, raw(xw), x, sz, size(xw)[end], L, tb, nt, smarg($smval), 1, qmfvec(wt))
            return x
        end
        function iswpdall(xw::HIP{T}, wt::OrthoFilter, arg::$AT, $(smsig...)) where T<:FT
            @assert 3 ≤ ndims(xw) ≤ 4
            sz = size(xw)[1:end-2]; N = size(xw)[end]
            x = newlike(xw, T, (sz..., N))
            L, tb, nt = treearg(arg)
            c_iswpd(T, raw(xw), x, sz, size(xw)[end-1], L, tb, nt, smarg($smval), N, qmfvec(wt))
            return x
        end
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# inverse autocorrelation transforms -- ACWT.jl, acwt/acwt_all.jl (1-D and 2-D, Float64)
# ---------------------------------------------------------------------------------------------------------------------
# iacdwt / iacwpt need no filter: v = (w₁ + w₂)/√2 (ACWT.jl:244-329, 537-648; acwt_all.jl:86-101, 189-204)
for (f, f!, fall, levels, cfun) in ((:iacdwt, :iacdwt!, :iacdwtall, :sdwt_levels, :c_iacdwt),
                                    (:iacwpt, :iacwpt!, :iacwptall, :swpt_levels, :c_iacwpt))
    for N in 1:2
        @eval function $f!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT
            @assert size(x) == size(xw)[1:end-1]
            $cfun(T, raw(xw), raw(x), size(x), $levels(size(x), size(xw)[end]), 1)
            return x
        end
    end
    @eval begin
        function $f(xw::HIP{T}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT
            @assert 2 ≤ ndims(xw) ≤ 3
            sz = size(xw)[1:end-1]
            x = newlike(xw, T, sz)
            $cfun(T, raw(xw), x, sz, $levels(sz, size(xw)[end]), 1)
            return x
        end
        function $fall(xw::HIP{T}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT
            @assert 3 ≤ ndims(xw) ≤ 4
            sz = size(xw)[1:end-2]; N = size(xw)[end]
            x = newlike(xw, T, (sz..., N))
            $cfun(T, raw(xw), x, sz, $levels(sz, size(xw)[end-1]), N)
            return x
        end
    end
end
# iacwpd by level or tree, with or without the (unused) filter argument (ACWT.jl:845-1000, acwt_all.jl:300-333)
iacwpd(xw::HIP{T}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT = iacwpd(xw, maxtransformlevels(xw, 1))
iacwpdall(xw::HIP{T}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT = iacwpdall(xw, maxtransformlevels(xw, 1))
for N in 1:2
    @eval iacwpd!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::Union{OrthoFilter,Nothing} = nothing) where T<:FT =
        iacwpd!(x, xw, maxtransformlevels(x))
end
for AT in (:Integer, :BitVector)
    for N in 1:2
        @eval begin
            function iacwpd!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, arg::$AT) where T<:FT
                @assert size(x) == size(xw)[1:end-1]
                L, tb, nt = treearg(arg)
                c_iacwpd(T, raw(xw), raw(x), size(x), size(xw)[end], L, tb, nt, 1)
                return x
            end
            iacwpd!(x::HIP{T,$N}, xw::AbstractArray{T,$(N + 1)}, wt::Union{OrthoFilter,Nothing}, arg::$AT) where T<:FT =
                iacwpd!(x, xw, arg)
        end
    end
    @eval begin
        function iacwpd(xw::HIP{T}, arg::$AT) where T<:FT
            @assert 2 ≤ ndims(xw) ≤ 3
            sz = size(xw)[1:end-1]
            x = newlike(xw, T, sz)
            L, tb, nt = treearg(arg)
            c_iacwpd(T, raw(xw), x, sz, size(xw)[end], L, tb, nt, 1)
            return x
        end
        iacwpd(xw::HIP{T}, wt::Union{OrthoFilter,Nothing}, arg::$AT) where T<:FT = iacwpd(xw, arg)
        function iacwpdall(xw::HIP{T}, arg::$AT) where T<:FT
            @assert 3 ≤ ndims(xw) ≤ 4
            sz = size(xw)[1:end-2]; N = size(xw)[end]
            x = newlike(xw, T, (sz..., N))
            L, tb, nt = treearg(arg)
            c_iacwpd(T, raw(xw), x, sz, size(xw)[end-1], L, tb, nt, N)
            return x
        end
        iacwpdall(xw::HIP{T}, wt::Union{OrthoFilter,Nothing}, arg::$AT) where T<:FT = iacwpdall(xw, arg)
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# best basis -- bestbasis/bestbasis_tree.jl:150-258, BestBasis.jl:59-110, 194-262
# ---------------------------------------------------------------------------------------------------------------------
costkind(c::LoglpCost) = (0, Float64(c.p))
costkind(c::NormCost) = (1, Float64(c.p))
bbkind(::ShannonEntropyCost) = 0
bbkind(::LogEnergyEntropyCost) = 1
jbb_ncost(sig, k, redundant) = redundant ? k : (length(sig) == 1 ? (1 << k) - 1 : gettreelength(1 << k, 1 << k))

"Σx and Σx² over the signal (last) axis of a decomposition (sz..., k, N); `into = (s, q)` adds to existing sums"
function jbb_moments(X::HIP{T}; into = nothing) where T<:FT
    shp = size(X)[1:end-1]; N = size(X)[end]
    s, q2 = into === nothing ? (newlike(X, T, shp), newlike(X, T, shp)) : into
    check(wx_jbb_moments(T, raw(X), s, q2, prod(shp), N, into === nothing ? 0 : 1, stream()))
    return s, q2
end
"tree_costs from the moments of `Ntot` signals (all ranks' sums after `allreduce_moments!`)"
function costs_from_moments(s::AbstractArray{T}, q2::AbstractArray{T}, Ntot::Integer, method::JBB) where T<:FT
    sig = size(s)[1:end-1]; k = size(s)[end]
    kind, p = costkind(method.cost)
    costs = Vector{T}(undef, jbb_ncost(sig, k, method.redundant))
    c_jbb_costs(T, s, q2, Ntot, sig, k, method.redundant, kind, p, costs)
    @assert !any(isnan, costs)        # the reference's `@assert all(σ .≥ 0)` (bestbasis_tree.jl:158, 189)
    return costs
end
for N in 1:2
    @eval begin
        # tree_costs(X, JBB) for 1-D (n, k, N) and 2-D (n, m, k, N) decompositions (bestbasis_tree.jl:150-207)
        function tree_costs(X::HIP{T,$(N + 2)}, method::JBB) where T<:FT
            s, q2 = jbb_moments(X)
            return costs_from_moments(s, q2, size(X)[end], method)
        end
        # tree_costs(X, BB) of one decomposed signal (n, k) / (n, m, k) (bestbasis_tree.jl:210-258)
        function tree_costs(X::HIP{T,$(N + 1)}, method::BB) where T<:FT
            sig = size(X)[1:end-1]; k = size(X)[end]
            costs = Vector{T}(undef, jbb_ncost(sig, k, method.redundant))
            c_bb_costs(T, raw(X), costs, sig, k, 1, method.redundant, bbkind(method.cost))
            return costs
        end
    end
end
"bestbasis_treeselection (BestBasis.jl:59-110) by the library's host routine; `costs` is mutated like the reference's"
function treeselect!(costs::Vector{T}, sig::Tuple, type::Symbol = :min) where T<:FT
    type in (:min, :max) || throw(ArgumentError("Unsupported type $type."))
    tree = Vector{UInt8}(undef, treelen(sig))
    c_treeselect(T, costs, length(costs), sig, type === :max ? 1 : 0, tree)
    return BitVector(tree .!= 0)
end
"the same selection for 1-D signals plus the margin of its closest decision, min |cc - pc| / |pc| (a tree is reproducible across
summation orders only while this is far above the rounding of the costs, ~1e-13): `(tree, min_rel_gap)`"
function treeselect_gap!(costs::Vector{T}, n::Integer, type::Symbol = :min) where T<:FT
    type in (:min, :max) || throw(ArgumentError("Unsupported type $type."))
    tree = Vector{UInt8}(undef, treelen((Int(n),)))
    gap = Ref{Float64}(Inf)
    check(wx_treeselect_gap(T, costs, length(costs), Int(n), type === :max ? 1 : 0, tree, gap))
    return BitVector(tree .!= 0), gap[]
end
bestbasis_treeselection(costs::HIP{T,1}, n::Integer, type::Symbol = :min) where T<:FT = treeselect!(raw(costs), (Int(n),), type)
bestbasis_treeselection(costs::HIP{T,1}, n::Integer, m::Integer, type::Symbol = :min) where T<:FT =
    treeselect!(raw(costs), (Int(n), Int(m)), type)
# bestbasistree(X, JBB) (BestBasis.jl:194-201), bestbasistree(X, BB) of one signal (:203-210)
bestbasistree(X::HIP{T}, method::JBB = JBB()) where T<:FT = treeselect!(tree_costs(X, method), size(X)[1:end-2])
bestbasistree(X::HIP{T}, method::BB) where T<:FT = treeselect!(tree_costs(X, method), size(X)[1:end-1])
# bestbasistreeall(X, BB) (BestBasis.jl:253-262): the costs of every signal and all N trees in one launch each
function bestbasistreeall(X::HIP{T}, method::BB) where T<:FT
    @assert 3 ≤ ndims(X) ≤ 4
    sig = size(X)[1:end-2]; k = size(X)[end-1]; N = size(X)[end]
    ncost = jbb_ncost(sig, k, method.redundant)
    costs = Matrix{T}(undef, ncost, N)
    c_bb_costs(T, raw(X), costs, sig, k, N, method.redundant, bbkind(method.cost))
    trees = Matrix{UInt8}(undef, treelen(sig), N)
    check(wx_treeselect_batch(T, costs, ncost, sig[1], length(sig) == 2 ? sig[2] : 0, 0, N, trees, stream()))
    return BitMatrix(trees .!= 0)
end

# acwpdall + JBB fused (BASELINE config 5): what `bestbasistree(acwpdall(x, wt, L), JBB(redundant = true))` returns,
# without the (n, 2^(L+1)-1, N) table -- 16 TiB for 262144 signals of 2048 samples.  The library walks the batch in
# chunks (depths ≤ 6 through a bounded table, the rest of the tree in registers), moments accumulate in signal order.
function acwpd_jbb_moments(x::HIP{Float64,2}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)); into = nothing)
    n, N = size(x); ncols = (1 << (L + 1)) - 1
    s, q2 = into === nothing ? (newlike(x, Float64, (n, ncols)), newlike(x, Float64, (n, ncols))) : into
    q = qmfvec(wt)
    check(wx_acwpd_jbb_moments(Float64, raw(x), s, q2, n, L, N, q, length(q), into === nothing ? 0 : 1, stream()))
    return s, q2
end
function tree_costs(x::HIP{Float64,2}, wt::OrthoFilter, L::Integer, method::JBB)
    method.redundant || throw(ArgumentError("acwpd is a redundant transform: pass JBB(redundant = true)"))
    s, q2 = acwpd_jbb_moments(x, wt, L)
    return costs_from_moments(s, q2, size(x, 2), method)
end
bestbasistree(x::HIP{Float64,2}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)), method::JBB = JBB(redundant = true)) =
    treeselect!(tree_costs(x, wt, L, method), (size(x, 1),))
"`(tree, min_rel_gap)` of the fused config-5 path"
bestbasistree_gap(x::HIP{Float64,2}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)), method::JBB = JBB(redundant = true)) =
    treeselect_gap!(tree_costs(x, wt, L, method), size(x, 1))

# ---------------------------------------------------------------------------------------------------------------------
# denoising -- Denoising.jl:146-166, 214-232, 285-327, 483-712
# ---------------------------------------------------------------------------------------------------------------------
thkind(::HardTH) = 0; thkind(::SoftTH) = 1; thkind(::SemiSoftTH) = 2; thkind(::SteinTH) = 3
colmask(::Nothing, k) = C_NULL
colmask(tree::BitVector, k) = Vector{UInt8}(getleaf(tree, :binary)[1:k])      # leaf columns of a swpd / acwpd table

"(first row, column) of the finest detail coefficients, 0-based, as noisest picks them (Denoising.jl:221-230)"
function detailrange(n, k, redundant::Bool, tree)
    redundant || return (tree === nothing ? n >> 1 : first(finestdetailrange(n, tree)) - 1, 0)
    return (0, tree === nothing ? k - 1 : finestdetailrange(n, tree, true)[2] - 1)
end
"noisest of every signal of a batch: x (n, N) dwt / wpt coefficients, or (n, k, N) redundant tables"
function noisestall(x::HIP{T}, redundant::Bool, tree::Union{BitVector,Nothing} = nothing) where T<:FT
    n = size(x, 1); N = size(x)[end]; k = ndims(x) == 2 ? 1 : size(x, 2)
    lo, col = detailrange(n, k, redundant, tree)
    sigma = Vector{T}(undef, N)
    check(wx_noisest(T, raw(x), n, k, N, lo, col, sigma, stream()))
    return sigma
end
# noisest(x, redundant[, tree]) of one decomposed signal (Denoising.jl:214-232)
function noisest(x::HIP{T}, redundant::Bool, tree::Union{BitVector,Nothing} = nothing) where T<:FT
    n = size(x, 1); k = ndims(x) == 1 ? 1 : size(x, 2)
    lo, col = detailrange(n, k, redundant, tree)
    sigma = Vector{T}(undef, 1)
    check(wx_noisest(T, raw(x), n, k, 1, lo, col, sigma, stream()))
    return sigma[1]
end
"threshold!(x, th, t) on rows row_lo+1:n of the selected columns of every signal; t: one value or one per signal"
function thresholdall!(x::HIP{T}, th::THType, t::AbstractVector{T}; row_lo::Integer = 0, cols = nothing) where T<:FT
    n = size(x, 1); N = size(x)[end]; k = ndims(x) == 2 ? 1 : size(x, 2)
    cm = cols === nothing ? C_NULL : Vector{UInt8}(cols)
    check(wx_threshold(T, raw(x), raw(x), n, k, N, thkind(th), t, length(t), row_lo, cm, stream()))
    return x
end
"iwptall(threshold(xw, th, scale .* t), wt, tree) in one pass over the coefficients (the threshold rides on the loads)"
function iwptall_thresholded(xw::HIP{T,2}, wt::OrthoFilter, tree::BitVector, th::THType, t::AbstractVector{T};
                             row_lo::Integer = 0, scale::Real = 1.0) where T<:FT
    n, N = size(xw)
    x̂ = newlike(xw, T, (n, N)); q = qmfvec(wt)
    _, tb, nt = treearg(tree)
    check(wx_iwpt1d_thresh(T, raw(xw), x̂, n, 0, tb, nt, N, q, length(q), thkind(th), t, length(t), row_lo, Float64(scale), stream()))
    return x̂
end
"idwtall(threshold(dwtall(x, wt, L), th, noisest.(columns) .* t), wt, L): denoiseall(x, :sig, wt; L, dnt, smooth) with estnoise = noisest"
function denoiseall_sig(x::HIP{T,2}, wt::OrthoFilter, L::Integer, th::THType, t::Real, undersmooth::Bool; coefs::Bool = false) where T<:FT
    n, N = size(x)
    x̂ = newlike(x, T, (n, N)); q = qmfvec(wt)
    if coefs       # x = dwtall(signals, wt, L): denoiseall(x, :dwt, wt; ...)
        check(wx_denoiseall_dwt(T, raw(x), x̂, n, L, N, q, length(q), thkind(th), Float64(t), undersmooth ? 1 : 0, C_NULL, stream()))
    else
        check(wx_denoiseall_sig(T, raw(x), x̂, n, L, N, q, length(q), thkind(th), Float64(t), undersmooth ? 1 : 0, C_NULL, stream()))
    end
    return x̂
end
# surethreshold / relerrorthreshold of one signal (Denoising.jl:146-166, 285-327) and of every signal of a batch (what
# SureShrink(xw, redundant, tree) and denoiseall(...; estnoise = relerrorthreshold) evaluate signal by signal)
function surethresholdall(coef::HIP{T}, redundant::Bool, tree::Union{BitVector,Nothing} = nothing; batched::Bool = true) where T<:FT
    n = size(coef, 1); N = batched ? size(coef)[end] : 1
    k = ndims(coef) - batched == 1 ? 1 : size(coef, 2)
    t = Vector{T}(undef, N)
    check(wx_surethreshold(T, raw(coef), n, k, N, colmask(redundant ? tree : nothing, k), t, stream()))
    return t
end
function relerrorthresholdall(coef::HIP{T}, redundant::Bool = false, tree::Union{BitVector,Nothing} = nothing,
                              elbows::Integer = 2; batched::Bool = true) where T<:FT
    @assert elbows ≥ 1
    n = size(coef, 1); N = batched ? size(coef)[end] : 1
    k = ndims(coef) - batched == 1 ? 1 : size(coef, 2)
    t = Vector{T}(undef, N)
    check(wx_relerrorthreshold(T, raw(coef), n, k, N, colmask(redundant ? tree : nothing, k), elbows, t, stream()))
    return t
end
surethreshold(coef::HIP{T}, redundant::Bool, tree::Union{BitVector,Nothing} = nothing) where T<:FT =
    surethresholdall(coef, redundant, tree; batched = false)[1]
relerrorthreshold(coef::HIP{T}, redundant::Bool = false, tree::Union{BitVector,Nothing} = nothing, elbows::Integer = 2) where T<:FT =
    relerrorthresholdall(coef, redundant, tree, elbows; batched = false)[1]

# denoiseall(x, :sig | :dwt | :wpt, wt; ...) (Denoising.jl:651-712) for the VisuShrink family as one device pipeline:
# transform, MAD of every signal, threshold riding on the inverse's loads.  The redundant input types compose from
# noisestall / thresholdall! and the inverse batch methods above exactly as waveletsext.jl_amd/denoising.py does.
function WaveletsExt.Denoising.denoiseall(x::HIP{T,2}, inputtype::Symbol, wt::OrthoFilter;
        L::Integer = maxtransformlevels(size(x, 1)), tree::BitVector = maketree(size(x, 1), L, :dwt),
        dnt = VisuShrink(size(x, 1)), estnoise::Union{Function,Vector{<:Number}} = noisest,
        bestTH::Union{Function,Nothing} = nothing, smooth::Symbol = :regular) where T<:FT
    @assert smooth in (:undersmooth, :regular)
    inputtype in (:sig, :dwt, :wpt) || throw(ArgumentError("device pipeline: inputtype :sig, :dwt or :wpt (compose the redundant types from noisestall / thresholdall!)"))
    estnoise isa Function && estnoise !== noisest && throw(ArgumentError("device pipeline: estnoise = noisest or precomputed values"))
    n = size(x, 1)
    if inputtype in (:sig, :dwt) && estnoise === noisest && bestTH === nothing
        # the whole pipeline behind one entry point (one pass over the signals / coefficients where the lattice kernel applies)
        return denoiseall_sig(x, wt, L, dnt.th, dnt.t, smooth === :undersmooth; coefs = inputtype === :dwt)
    end
    xw = inputtype === :sig ? HIP(dwtall(x, wt, L)) : x
    tr = inputtype === :wpt ? tree : maketree(n, L, :dwt)
    # σᵢ = estnoise(xᵢ, false, tree | nothing) (Denoising.jl:683-686), or the caller's precomputed values
    σ = estnoise isa Function ? noisestall(xw, false, inputtype === :wpt ? tree : nothing) : Vector{T}(estnoise)
    bestTH === nothing || (σ = fill(T(bestTH(σ)), length(σ)))               # summary threshold (Denoising.jl:697-700)
    lo = smooth === :undersmooth ? (inputtype === :wpt ? last(coarsestscalingrange(n, tree)) : nodelength(n, L)) : 0
    return iwptall_thresholded(xw, wt, tr, dnt.th, σ; row_lo = lo, scale = dnt.t)
end

# ---------------------------------------------------------------------------------------------------------------------
# Local Discriminant Basis, the batch-sized steps -- ldb/ldb_energymap.jl:109-238, ldb/ldb_measures.jl:185-201, 441-519
# ---------------------------------------------------------------------------------------------------------------------
"class index (0-based, unique(y) order) of every signal, and the number of classes"
function classindex(y)
    c = unique(y)
    return Int32[findfirst(==(v), c) - 1 for v in y], length(c)
end
# energy_map(Xw, y, TimeFrequency()) (ldb_energymap.jl:109-141): Γ (sz..., k, nc)
function energy_map(Xw::HIP{T}, y::AbstractVector, ::TimeFrequency) where T<:FT
    nd = ndims(Xw)
    @assert 3 ≤ nd ≤ 4
    cls, nc = classindex(y)
    sz = size(Xw)[1:nd-2]; k = size(Xw, nd - 1); N = size(Xw, nd)
    @assert N == length(y)
    @assert nc > 1
    @assert 1 ≤ k - 1 ≤ maxtransformlevels(min(sz...))
    Γ = newlike(Xw, T, (sz..., k, nc))
    check(wx_energy_map(T, raw(Xw), prod(sz) * k, prod(sz), N, cls, nc, Γ, C_NULL, stream()))
    return Γ
end
# energy_map(Xw, y, ProbabilityDensity()) (ldb_energymap.jl:143-184): average shifted histograms per coefficient and class
function energy_map(Xw::HIP{T}, y::AbstractVector, ::ProbabilityDensity) where T<:FT
    cls, nc = classindex(y); N = size(Xw)[end]; ne = length(Xw) ÷ N
    nbins = ceil(Int, (30 * N)^(1 / 5)); plen = (nbins + 1) * ceil(Int, 100 / nbins)
    Γ = newlike(Xw, Float64, (size(Xw)[1:end-1]..., plen, nc))
    check(wx_pdf_energy_map(T, raw(Xw), ne, N, cls, nc, Γ, stream()))
    return Γ
end
"the :pdf weights of energy_map(Xw, y, Signatures(:pdf)) (ldb_energymap.jl:216-232), one per coefficient and signal"
function signature_weights(Xw::HIP{T}, y::AbstractVector) where T<:FT
    cls, nc = classindex(y); N = size(Xw)[end]; ne = length(Xw) ÷ N
    W = newlike(Xw, T, size(Xw))
    check(wx_signature_weights(T, raw(Xw), ne, N, cls, nc, W, stream()))
    return W
end
"discriminant_measure(energy_map(Xw, y, Signatures(w)), EarthMoverDistance()) per coefficient (ldb_measures.jl:185-201, 254-360)"
function emd_measure(Xw::HIP{T}, y::AbstractVector, W::Union{AbstractArray{T},Nothing} = nothing) where T<:FT
    cls, nc = classindex(y); N = size(Xw)[end]; ne = length(Xw) ÷ N
    D = newlike(Xw, T, size(Xw)[1:end-1])
    if W === nothing
        check(wx_emd_measure(T, raw(Xw), ne, N, cls, nc, D, stream()))
    else
        check(wx_emd_measure_weighted(T, raw(Xw), raw(W), ne, N, cls, nc, D, stream()))
    end
    return D
end
"per-class mean and variance over the signal axis (n-1 denominator): the tables of FishersClassSeparability"
function class_mean_var(coefs::HIP{T}, y::AbstractVector) where T<:FT
    cls, nc = classindex(y); N = size(coefs)[end]; ne = length(coefs) ÷ N
    μ = newlike(coefs, T, (size(coefs)[1:end-1]..., nc)); v = similar(μ)
    check(wx_class_mean(T, raw(coefs), ne, N, cls, nc, μ, stream()))
    check(wx_class_var(T, raw(coefs), ne, N, cls, nc, μ, v, stream()))
    return μ, v, cls, nc
end
"per-class median and mad(normalize = false) over the signal axis: the tables of RobustFishersClassSeparability"
function class_median_mad(coefs::HIP{T}, y::AbstractVector) where T<:FT
    cls, nc = classindex(y); N = size(coefs)[end]; ne = length(coefs) ÷ N
    med = newlike(coefs, T, (size(coefs)[1:end-1]..., nc)); md = similar(med)
    check(wx_class_median_mad(T, raw(coefs), ne, N, cls, nc, med, md, stream()))
    return med, md, cls, nc
end
# discriminant_power on the small class tables (ldb_measures.jl:441-519); `centre` is the mean / median over classes
function power_from_tables(loc::AbstractArray{T}, spread::AbstractArray{T}, centre, cls, nc) where T
    nd = ndims(loc)
    pᵢ = reshape(T[count(==(c), cls) for c in 0:nc-1] ./ T(length(cls)), ntuple(_ -> 1, nd - 1)..., nc)
    power = dropdims(sum((loc .- centre .* loc) .^ 2 .* pᵢ, dims = nd) ./ sum(spread .* pᵢ, dims = nd), dims = nd)
    return power, sortperm(vec(power), rev = true)
end
function discriminant_power(coefs::HIP{T}, y::AbstractVector, ::FishersClassSeparability) where T<:FT
    @assert 2 ≤ ndims(coefs) ≤ 3
    μ, v, cls, nc = class_mean_var(coefs, y)
    μh, vh = Array(μ), Array(v)
    return power_from_tables(μh, vh, mean(μh, dims = ndims(μh)), cls, nc)
end
function discriminant_power(coefs::HIP{T}, y::AbstractVector, ::RobustFishersClassSeparability) where T<:FT
    @assert 2 ≤ ndims(coefs) ≤ 3
    med, md, cls, nc = class_median_mad(coefs, y)
    mh, dh = Array(med), Array(md)
    return power_from_tables(mh, dh, median(mh, dims = ndims(mh)), cls, nc)
end

# ---------------------------------------------------------------------------------------------------------------------
# shift-invariant packet decomposition for a whole batch (SIWT.jl:57-229 per signal)
# ---------------------------------------------------------------------------------------------------------------------
# The flat table of include/waveletsext_hip.h instead of one Dict of node objects per signal: W (n, NS, N), node
# (j, i, t) = W[i*(n>>j)+1 : (i+1)*(n>>j), coloff(j) + (t >> max(0, j-d)) + 1, signal].  `siwt_node` rebuilds the
# reference's ShiftInvariantWaveletTransformNode for callers that want the object model back.
struct SIWTBatch{T}
    W::Array{T,3}; costs::Matrix{T}; status::Matrix{UInt8}; wt::OrthoFilter; L::Int; d::Int
end
siwt_coloff(j, d) = sum(1 << min(i, d) for i in 0:j-1; init = 0)
siwt_nodeoff(j, d) = sum((1 << min(i, d)) << i for i in 0:j-1; init = 0)
function siwpdall(x::HIP{T,2}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)), d::Integer = L) where T<:FT
    n, N = size(x)
    NS = wx_siwt_ncols(L, d); NN = wx_siwt_nnodes(L, d)
    W = Array{T,3}(undef, n, NS, N); costs = Matrix{T}(undef, NN, N)
    q = qmfvec(wt)
    check(wx_siwpd(T, raw(x), W, costs, n, L, d, N, q, length(q), stream()))
    return SIWTBatch{T}(W, costs, fill(0x01, NN, N), wt, L, d)
end
function bestbasistreeall!(b::SIWTBatch{T}) where T<:FT                      # siwt/siwt_bestbasis.jl:28-102 per signal
    check(wx_siwt_bestbasis(T, b.costs, b.status, b.L, b.d, size(b.W, 3), stream()))
    return b.status
end
function isiwpdall(b::SIWTBatch{T}; literal::Bool = false) where T<:FT       # literal: the flag as siwt_one_level.jl:126 spells it
    n, _, N = size(b.W)
    xh = Matrix{T}(undef, n, N); q = qmfvec(b.wt)
    check(wx_isiwpd(T, b.W, b.status, xh, n, b.L, b.d, N, q, length(q), literal, stream()))
    return xh
end
function siwt_node(b::SIWTBatch{T}, sig::Integer, j::Integer, i::Integer, t::Integer) where T
    n = size(b.W, 1); m = max(0, j - b.d); slot = t >> m; len = n >> j
    v = b.W[i*len+1:(i+1)*len, siwt_coloff(j, b.d) + slot + 1, sig]
    return WaveletsExt.SIWT.ShiftInvariantWaveletTransformNode{1,Int,T}(j, i, t,
        b.costs[siwt_nodeoff(j, b.d) + (slot << j) + i + 1, sig], v)
end

# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU (one process per GPU; include/waveletsext_hip.h "Multi-GPU exchange")
# ---------------------------------------------------------------------------------------------------------------------
# The launcher (MPI.jl, Distributed.jl) broadcasts the 128-byte id made on rank 0.  Transforms need no collective: each
# process runs the methods above on its contiguous shard `x[:, lo:hi]`.  The buffers of the two exchange steps are
# device arrays (ROCArray) or device pointers, `count` elements per rank.
struct Comm; handle::Ptr{Cvoid}; nranks::Int; rank::Int; end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(wx_comm_unique_id(id))
    return id
end
function Comm(nranks::Integer, rank::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(wx_comm_init(nranks, rank, id, h))
    return Comm(h[], nranks, rank)
end
Base.close(c::Comm) = check(wx_comm_destroy(c.handle))
# C1: reconstructed output shards -> full batch on every rank (recv holds nranks*count elements)
allgather_out!(::Type{T}, recv, send, count::Integer, c::Comm) where T<:FT =
    check(wx_allgather_out(T, send, recv, count, c.handle, stream()))
# C1 for ragged shards (B mod nranks != 0): counts[r] = elements of rank r's shard (signal length x its share of the batch);
# every shard lands at its offset of recv on every rank, no padding
function allgatherv_out!(::Type{T}, recv, send, counts::AbstractVector{<:Integer}, c::Comm) where T<:FT
    length(counts) == c.nranks || throw(ArgumentError("counts needs one entry per rank"))
    check(wx_allgatherv_out(T, send, recv, Vector{Int64}(counts), c.nranks, c.handle, stream()))
end
# C2: JBB moments [Σx | Σx²] summed over ranks in place, then costs_from_moments / treeselect! on every rank
allreduce_moments!(::Type{T}, buf, count::Integer, c::Comm) where T<:FT =
    check(wx_allreduce_moments(T, buf, count, c.handle, stream()))

end # module
