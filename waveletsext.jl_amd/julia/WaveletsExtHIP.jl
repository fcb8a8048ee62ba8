# WaveletsExtHIP.jl -- reference-side binding for libwaveletsext_hip.so (delivered as source: there is no
# Julia in the build image or on the GPU box, so this file is NOT exercised by the test suite; every
# entry point it calls is exercised through the same C ABI by tests/ via ctypes).
#
# It adds methods to the generic functions of Wavelets.jl / WaveletsExt.jl for a marker wrapper type
# `HIP(x)`, so existing user code switches the hot path by wrapping its input:
#
#     using Wavelets, WaveletsExt, WaveletsExtHIP
#     xw = wpdall(HIP(x), wt, L)            # MI355X kernels instead of dwt/dwt_all.jl:260-282
#     x̂  = iwpdall(HIP(xw), wt, tree)
#     tr = bestbasistree(HIP(acwpdall(HIP(x), wt)), JBB(redundant=true))
#
# Plain `Array`s are passed as host pointers (the library stages H2D/D2H); AMDGPU.jl `ROCArray`s can be
# passed unchanged (device pointers, asynchronous on `stream`).
module WaveletsExtHIP

using Wavelets, WaveletsExt
import Wavelets.Transforms: wpt, wpt!, iwpt, iwpt!
import Wavelets.Threshold: bestbasistree, HardTH, SoftTH, SemiSoftTH, SteinTH
import WaveletsExt.DWT: wpd, wpd!, iwpd, iwpd!, wpdall, iwpdall, wptall, iwptall
import WaveletsExt.SWT: sdwtall, isdwtall, swptall, iswptall, swpdall, iswpdall
import WaveletsExt.ACWT: acdwtall, iacdwtall, acwptall, iacwptall, acwpdall, iacwpdall
import WaveletsExt.BestBasis: tree_costs, JBB, LoglpCost, NormCost, bestbasis_treeselection, BB, ShannonEntropyCost,
                             LogEnergyEntropyCost, bestbasistreeall

export HIP

const LIB = get(ENV, "WAVELETSEXT_HIP_LIB", "libwaveletsext_hip.so")

"Marker wrapper: dispatches the hot path to the HIP library."
struct HIP{T,N,A<:AbstractArray{T,N}} <: AbstractArray{T,N}
    a::A
end
Base.size(x::HIP) = size(x.a)
Base.getindex(x::HIP, i...) = getindex(x.a, i...)
Base.parent(x::HIP) = x.a

const WX_EASSERT, WX_EARG, WX_EBOUNDS = Cint(-1), Cint(-2), Cint(-3)

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:wx_last_error, LIB), Cstring, ()))
    rc == WX_EASSERT && throw(AssertionError(msg))
    rc == WX_EARG && throw(ArgumentError(msg))
    rc == WX_EBOUNDS && throw(BoundsError())
    error("libwaveletsext_hip status $rc: $msg")
end

sfx(::Type{Float64}) = "_f64"
sfx(::Type{Float32}) = "_f32"
treebytes(tree::BitVector) = Vector{UInt8}(tree)
qmfvec(wt::OrthoFilter) = Vector{Float64}(WT.qmf(wt))
batchof(x, nsig) = prod(size(x)[(nsig+1):end])

# one @eval per element type keeps the ccall symbol a compile-time constant
for (T, S) in ((Float64, "_f64"), (Float32, "_f32"))
    @eval begin
        # ---- wpdall / wpd (dwt/dwt_all.jl:260-282, DWT.jl:131-209) -------------------------------------------
        function wpdall(x::HIP{$T}, wt::OrthoFilter, L::Integer = maxtransformlevels(minimum(size(x)[1:end-1])))
            @assert ndims(x) > 1
            sz = size(x)[1:end-1]; N = size(x)[end]
            y = Array{$T}(undef, (sz..., L + 1, N)); q = qmfvec(wt)
            if length(sz) == 1
                check(ccall(($("wx_wpd1d" * S), LIB), Cint,
                            (Ptr{$T}, Ptr{$T}, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                            parent(x), y, sz[1], L, N, q, length(q), C_NULL))
            else
                check(ccall(($("wx_wpd2d" * S), LIB), Cint,
                            (Ptr{$T}, Ptr{$T}, Int64, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                            parent(x), y, sz[1], sz[2], L, N, q, length(q), C_NULL))
            end
            return y
        end

        # ---- iwpdall (dwt/dwt_all.jl:324-342), by level or by tree --------------------------------------------
        function iwpdall(xw::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(minimum(size(xw)[1:end-2])))
            @assert ndims(xw) > 2
            sz = size(xw)[1:end-2]; k = size(xw)[end-1]; N = size(xw)[end]
            x̂ = Array{$T}(undef, (sz..., N)); q = qmfvec(wt)
            L, tree = arg isa BitVector ? (0, treebytes(arg)) : (Int(arg), UInt8[])
            tp = isempty(tree) ? Ptr{UInt8}(C_NULL) : pointer(tree)
            GC.@preserve tree begin
                if length(sz) == 1
                    check(ccall(($("wx_iwpd1d" * S), LIB), Cint,
                                (Ptr{$T}, Ptr{$T}, Int64, Cint, Cint, Ptr{UInt8}, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                                parent(xw), x̂, sz[1], k, L, tp, length(tree), N, q, length(q), C_NULL))
                else
                    check(ccall(($("wx_iwpd2d" * S), LIB), Cint,
                                (Ptr{$T}, Ptr{$T}, Int64, Int64, Cint, Cint, Ptr{UInt8}, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                                parent(xw), x̂, sz[1], sz[2], k, L, tp, length(tree), N, q, length(q), C_NULL))
                end
            end
            return x̂
        end

        # ---- wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225) -----------------------------------------------
        function _wptall(sym1::Symbol, x::HIP{$T}, wt::OrthoFilter, arg)
            @assert ndims(x) > 1
            sz = size(x)[1:end-1]; N = size(x)[end]
            y = similar(parent(x)); q = qmfvec(wt)
            L, tree = arg isa BitVector ? (0, treebytes(arg)) : (Int(arg), UInt8[])
            tp = isempty(tree) ? Ptr{UInt8}(C_NULL) : pointer(tree)
            GC.@preserve tree begin
                if length(sz) == 1
                    f = sym1 === :fwd ? $("wx_wpt1d" * S) : $("wx_iwpt1d" * S)
                    check(ccall((f, LIB), Cint,
                                (Ptr{$T}, Ptr{$T}, Int64, Cint, Ptr{UInt8}, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                                parent(x), y, sz[1], L, tp, length(tree), N, q, length(q), C_NULL))
                else
                    f = sym1 === :fwd ? $("wx_wpt2d" * S) : $("wx_iwpt2d" * S)
                    check(ccall((f, LIB), Cint,
                                (Ptr{$T}, Ptr{$T}, Int64, Int64, Cint, Ptr{UInt8}, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                                parent(x), y, sz[1], sz[2], L, tp, length(tree), N, q, length(q), C_NULL))
                end
            end
            return y
        end
        wptall(x::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(minimum(size(x)[1:end-1]))) = _wptall(:fwd, x, wt, arg)
        iwptall(x::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(minimum(size(x)[1:end-1]))) = _wptall(:inv, x, wt, arg)
        # single-signal methods are the batch-1 case of the same entry points
        wpt(x::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(parent(x))) =
            dropdims(wptall(HIP(reshape(parent(x), size(x)..., 1)), wt, arg), dims = ndims(x) + 1)
        iwpt(x::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(parent(x))) =
            dropdims(iwptall(HIP(reshape(parent(x), size(x)..., 1)), wt, arg), dims = ndims(x) + 1)
        wpd(x::HIP{$T}, wt::OrthoFilter, L::Integer = maxtransformlevels(parent(x))) =
            dropdims(wpdall(HIP(reshape(parent(x), size(x)..., 1)), wt, L), dims = ndims(x) + 2)
        iwpd(xw::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(minimum(size(xw)[1:end-1]))) =
            dropdims(iwpdall(HIP(reshape(parent(xw), size(xw)..., 1)), wt, arg), dims = ndims(xw))

        # ---- stationary family (swt/swt_all.jl) ----------------------------------------------------------------
        function _swt_fwd(f, ncols, x::HIP{$T}, wt::OrthoFilter, L::Integer)
            @assert ndims(x) == 2
            n, N = size(x); q = qmfvec(wt)
            xw = Array{$T}(undef, (n, ncols, N))
            check(ccall((f, LIB), Cint, (Ptr{$T}, Ptr{$T}, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                        parent(x), xw, n, L, N, q, length(q), C_NULL))
            return xw
        end
        sdwtall(x::HIP{$T}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _swt_fwd($("wx_sdwt1d" * S), L + 1, x, wt, L)
        swptall(x::HIP{$T}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _swt_fwd($("wx_swpt1d" * S), 1 << L, x, wt, L)
        swpdall(x::HIP{$T}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _swt_fwd($("wx_swpd1d" * S), 1 << (L + 1) - 1, x, wt, L)

        function isdwtall(xw::HIP{$T}, wt::OrthoFilter, sm::Integer = -1)     # sm < 0: average-based (swt_all.jl:89)
            n, k, N = size(xw); x = Array{$T}(undef, (n, N)); q = qmfvec(wt)
            check(ccall(($("wx_isdwt1d" * S), LIB), Cint, (Ptr{$T}, Ptr{$T}, Int64, Cint, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                        parent(xw), x, n, k - 1, sm, N, q, length(q), C_NULL))
            return x
        end
        function iswptall(xw::HIP{$T}, wt::OrthoFilter, sm::Integer = -1)
            n, m, N = size(xw); x = Array{$T}(undef, (n, N)); q = qmfvec(wt)
            isdyadic(m) || throw(ArgumentError("Number of columns of xw is not dyadic."))
            check(ccall(($("wx_iswpt1d" * S), LIB), Cint, (Ptr{$T}, Ptr{$T}, Int64, Cint, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                        parent(xw), x, n, ndyadicscales(m), sm, N, q, length(q), C_NULL))
            return x
        end
        function iswpdall(xw::HIP{$T}, wt::OrthoFilter, arg = maxtransformlevels(size(xw, 1)), sm::Integer = -1)
            n, m, N = size(xw); x = Array{$T}(undef, (n, N)); q = qmfvec(wt)
            L, tree = arg isa BitVector ? (0, treebytes(arg)) : (Int(arg), UInt8[])
            tp = isempty(tree) ? Ptr{UInt8}(C_NULL) : pointer(tree)
            GC.@preserve tree check(ccall(($("wx_iswpd1d" * S), LIB), Cint,
                (Ptr{$T}, Ptr{$T}, Int64, Int64, Cint, Ptr{UInt8}, Int64, Int64, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                parent(xw), x, n, m, L, tp, length(tree), sm, N, q, length(q), C_NULL))
            return x
        end
    end
end

# ---- autocorrelation family (Float64 only, acwt/acwt_all.jl) ---------------------------------------------------
function _ac_fwd(f, ncols, x::HIP{Float64}, wt::OrthoFilter, L::Integer)
    n, N = size(x); q = qmfvec(wt)
    xw = Array{Float64}(undef, (n, ncols, N))
    check(ccall((f, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                parent(x), xw, n, L, N, q, length(q), C_NULL))
    return xw
end
acdwtall(x::HIP{Float64}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _ac_fwd(:wx_acdwt1d_f64, L + 1, x, wt, L)
acwptall(x::HIP{Float64}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _ac_fwd(:wx_acwpt1d_f64, 1 << L, x, wt, L)
acwpdall(x::HIP{Float64}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1))) = _ac_fwd(:wx_acwpd1d_f64, 1 << (L + 1) - 1, x, wt, L)

function iacdwtall(xw::HIP{Float64}, wt = nothing)
    n, k, N = size(xw); x = Array{Float64}(undef, (n, N))
    check(ccall((:wx_iacdwt1d_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Cint, Int64, Ptr{Cvoid}), parent(xw), x, n, k - 1, N, C_NULL))
    return x
end
function iacwptall(xw::HIP{Float64}, wt = nothing)
    n, m, N = size(xw); x = Array{Float64}(undef, (n, N))
    check(ccall((:wx_iacwpt1d_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Cint, Int64, Ptr{Cvoid}), parent(xw), x, n, ndyadicscales(m), N, C_NULL))
    return x
end
function iacwpdall(xw::HIP{Float64}, arg = maxtransformlevels(size(xw, 1)))
    n, m, N = size(xw); x = Array{Float64}(undef, (n, N))
    L, tree = arg isa BitVector ? (0, treebytes(arg)) : (Int(arg), UInt8[])
    tp = isempty(tree) ? Ptr{UInt8}(C_NULL) : pointer(tree)
    GC.@preserve tree check(ccall((:wx_iacwpd1d_f64, LIB), Cint,
        (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Cint, Ptr{UInt8}, Int64, Int64, Ptr{Cvoid}),
        parent(xw), x, n, m, L, tp, length(tree), N, C_NULL))
    return x
end
iacwpdall(xw::HIP{Float64}, wt::Union{OrthoFilter,Nothing}, arg) = iacwpdall(xw, arg)

# ---- JBB (bestbasis/bestbasis_tree.jl:150-180, BestBasis.jl:194-201) -------------------------------------------
costkind(c::LoglpCost) = (Cint(0), Float64(c.p))
costkind(c::NormCost) = (Cint(1), Float64(c.p))

function tree_costs(X::HIP{Float64,3}, method::JBB)
    n, k, N = size(X)
    s = Array{Float64}(undef, (n, k)); q = similar(s)
    check(ccall((:wx_jbb_moments_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Cint, Ptr{Cvoid}),
                parent(X), s, q, n * k, N, 0, C_NULL))
    kind, p = costkind(method.cost)
    costs = Vector{Float64}(undef, method.redundant ? k : 1 << k - 1)
    check(ccall((:wx_jbb_costs_f64, LIB), Cint,
                (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Cint, Cint, Float64, Ptr{Float64}, Ptr{Cvoid}),
                s, q, N, n, k, method.redundant, kind, p, costs, C_NULL))
    @assert !any(isnan, costs)        # the reference's `@assert all(σ .>= 0)` (bestbasis_tree.jl:158)
    return costs
end

function bestbasistree(X::HIP{Float64,3}, method::JBB = JBB())
    costs = tree_costs(X, method)
    n = size(X, 1)
    tree = Vector{UInt8}(undef, n - 1)
    check(ccall((:wx_treeselect_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Cint, Ptr{UInt8}), costs, length(costs), n, 0, tree))
    return BitVector(tree .!= 0)
end

# ---- standard best basis for a whole batch (BestBasis.jl:253-262): costs + all trees on the device -------------
bbkind(::ShannonEntropyCost) = Cint(0)
bbkind(::LogEnergyEntropyCost) = Cint(1)
function bestbasistreeall(X::HIP{Float64,3}, method::BB)
    n, k, N = size(X)
    ncost = method.redundant ? k : 1 << k - 1
    costs = Matrix{Float64}(undef, ncost, N)
    check(ccall((:wx_bb_costs_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Cint, Cint, Ptr{Cvoid}),
                parent(X), costs, n, k, N, method.redundant, bbkind(method.cost), C_NULL))
    trees = Matrix{UInt8}(undef, n - 1, N)
    check(ccall((:wx_treeselect_batch_f64, LIB), Cint,
                (Ptr{Float64}, Int64, Int64, Int64, Cint, Int64, Ptr{UInt8}, Ptr{Cvoid}), costs, ncost, n, 0, 0, N, trees, C_NULL))
    return BitMatrix(trees .!= 0)
end

# ---- denoising core (Denoising.jl:214-232, 651-712) for a batch of dwt-decomposed signals ------------------------
# sigma_i = mad!(finest details of signal i)/0.6745 and the threshold step on the device; the transforms are the
# batch methods above (`dwtall`/`idwtall` of WaveletsExt dispatch on HIP like `wptall`).  Other input types differ
# only in (row_lo, col, colmask), see waveletsext.jl_amd/denoising.py.
thkind(::HardTH) = Cint(0); thkind(::SoftTH) = Cint(1); thkind(::SemiSoftTH) = Cint(2); thkind(::SteinTH) = Cint(3)
function noisestall(xw::HIP{Float64,2})
    n, N = size(xw)
    sigma = Vector{Float64}(undef, N)
    check(ccall((:wx_noisest_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Cvoid}),
                parent(xw), n, 1, N, n >> 1, 0, sigma, C_NULL))
    return sigma
end
function thresholdall!(xw::HIP{Float64,2}, th, t::Vector{Float64}; row_lo::Integer = 0)
    n, N = size(xw)
    check(ccall((:wx_threshold_f64, LIB), Cint,
                (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Cint, Ptr{Float64}, Int64, Int64, Ptr{UInt8}, Ptr{Cvoid}),
                parent(xw), parent(xw), n, 1, N, thkind(th), t, length(t), row_lo, C_NULL, C_NULL))
    return xw
end

# threshold selection of SureShrink / RelErrorShrink for every signal (Denoising.jl:146-166, 285-327): what
# `SureShrink(xw, redundant, tree)` and `denoiseall(...; estnoise = relerrorthreshold)` evaluate signal by signal.
# `leaves` = getleaf(tree, :binary) for swpd / acwpd tables (nothing = every column).
function _colmask(leaves, k)
    leaves === nothing && return C_NULL
    return UInt8.(leaves[1:k])
end
function surethresholdall(xw::HIP{Float64}, leaves = nothing)
    n = size(xw, 1); N = size(xw, ndims(xw)); k = ndims(xw) == 2 ? 1 : size(xw, 2)
    t = Vector{Float64}(undef, N)
    check(ccall((:wx_surethreshold_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{UInt8}, Ptr{Float64}, Ptr{Cvoid}),
                parent(xw), n, k, N, _colmask(leaves, k), t, C_NULL))
    return t
end
function relerrorthresholdall(xw::HIP{Float64}, leaves = nothing, elbows::Integer = 2)
    n = size(xw, 1); N = size(xw, ndims(xw)); k = ndims(xw) == 2 ? 1 : size(xw, 2)
    t = Vector{Float64}(undef, N)
    check(ccall((:wx_relerrorthreshold_f64, LIB), Cint,
                (Ptr{Float64}, Int64, Int64, Int64, Ptr{UInt8}, Cint, Ptr{Float64}, Ptr{Cvoid}),
                parent(xw), n, k, N, _colmask(leaves, k), elbows, t, C_NULL))
    return t
end

# 3-D dwtall / idwtall (dwt_all.jl:39-54, 95-110 on 4-D arrays: cubes with dyadic sides)
function WaveletsExt.dwtall(x::HIP{Float64,4}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)))
    y = similar(x); q = WT.qmf(wt)
    check(ccall((:wx_dwt3d_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                parent(x), parent(y), size(x, 1), size(x, 2), size(x, 3), L, size(x, 4), q, length(q), C_NULL))
    return y
end
function WaveletsExt.idwtall(xw::HIP{Float64,4}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(xw, 1)))
    y = similar(xw); q = WT.qmf(wt)
    check(ccall((:wx_idwt3d_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                parent(xw), parent(y), size(xw, 1), size(xw, 2), size(xw, 3), L, size(xw, 4), q, length(q), C_NULL))
    return y
end

# LDB order statistics and density maps over the signal axis (ldb/ldb_measures.jl:185-201, 254-360, 481-519,
# ldb/ldb_energymap.jl:143-238).  cls[i] in 0:nc-1 = index of y[i] in unique(y).
function _cls(y)
    c = unique(y)
    return Int32[findfirst(==(v), c) - 1 for v in y], length(c)
end
function class_median_mad(coefs::HIP{Float64}, y)
    cls, nc = _cls(y); N = size(coefs, ndims(coefs)); ne = length(coefs) ÷ N
    med = Array{Float64}(undef, size(coefs)[1:end-1]..., nc); mad = similar(med)
    check(ccall((:wx_class_median_mad_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                parent(coefs), ne, N, cls, nc, med, mad, C_NULL))
    return med, mad                       # discriminant_power(coefs, y, RobustFishersClassSeparability()) finishes on these
end
function emd_measure(Xw::HIP{Float64}, y, W::Union{Nothing,HIP{Float64}} = nothing)   # discriminant_measure(energy_map(Xw, y, Signatures()), EarthMoverDistance())
    cls, nc = _cls(y); N = size(Xw, ndims(Xw)); ne = length(Xw) ÷ N
    D = Array{Float64}(undef, size(Xw)[1:end-1]...)
    if W === nothing
        check(ccall((:wx_emd_measure_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Cvoid}),
                    parent(Xw), ne, N, cls, nc, D, C_NULL))
    else
        check(ccall((:wx_emd_measure_weighted_f64, LIB), Cint,
                    (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Cvoid}),
                    parent(Xw), parent(W), ne, N, cls, nc, D, C_NULL))
    end
    return D
end
function WaveletsExt.energy_map(Xw::HIP{Float64}, y, ::ProbabilityDensity)
    cls, nc = _cls(y); N = size(Xw, ndims(Xw)); ne = length(Xw) ÷ N
    nbins = ceil(Int, (30 * N)^(1 / 5)); plen = (nbins + 1) * ceil(Int, 100 / nbins)
    G = Array{Float64}(undef, size(Xw)[1:end-1]..., plen, nc)
    check(ccall((:wx_pdf_energy_map_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Cvoid}),
                parent(Xw), ne, N, cls, nc, G, C_NULL))
    return G
end
function signature_weights(Xw::HIP{Float64}, y)          # the :pdf weights of energy_map(Xw, y, Signatures(:pdf))
    cls, nc = _cls(y); N = size(Xw, ndims(Xw)); ne = length(Xw) ÷ N
    W = similar(Xw)
    check(ccall((:wx_signature_weights_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Cvoid}),
                parent(Xw), ne, N, cls, nc, parent(W), C_NULL))
    return W
end

# ---- shift-invariant packet decomposition for a whole batch (SIWT.jl:57-229 per signal) --------------------------
# The flat table of include/waveletsext_hip.h instead of one Dict of node objects per signal: W (n, NS, N), node
# (j, i, t) = W[i*(n>>j)+1 : (i+1)*(n>>j), coloff(j) + (t >> max(0, j-d)) + 1, signal].  `siwt_node` rebuilds the
# reference's ShiftInvariantWaveletTransformNode for callers that want the object model back.
struct SIWTBatch
    W::Array{Float64,3}; costs::Matrix{Float64}; status::Matrix{UInt8}; wt::OrthoFilter; L::Int; d::Int
end
siwt_coloff(j, d) = sum(1 << min(i, d) for i in 0:j-1; init = 0)
siwt_nodeoff(j, d) = sum((1 << min(i, d)) << i for i in 0:j-1; init = 0)
function siwpdall(x::HIP{Float64,2}, wt::OrthoFilter, L::Integer = maxtransformlevels(size(x, 1)), d::Integer = L)
    n, N = size(x)
    NS = ccall((:wx_siwt_ncols, LIB), Int64, (Cint, Cint), L, d)
    NN = ccall((:wx_siwt_nnodes, LIB), Int64, (Cint, Cint), L, d)
    W = Array{Float64,3}(undef, n, NS, N); costs = Matrix{Float64}(undef, NN, N)
    q = qmfvec(wt)
    check(ccall((:wx_siwpd_f64, LIB), Cint,
                (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Int64, Ptr{Float64}, Cint, Ptr{Cvoid}),
                parent(x), W, costs, n, L, d, N, q, length(q), C_NULL))
    return SIWTBatch(W, costs, fill(0x01, NN, N), wt, L, d)
end
function bestbasistreeall!(b::SIWTBatch)
    check(ccall((:wx_siwt_bestbasis_f64, LIB), Cint, (Ptr{Float64}, Ptr{UInt8}, Cint, Cint, Int64, Ptr{Cvoid}),
                b.costs, b.status, b.L, b.d, size(b.W, 3), C_NULL))
    return b.status
end
function isiwpdall(b::SIWTBatch; literal::Bool = false)   # literal: the flag as siwt_one_level.jl:126 spells it
    n, _, N = size(b.W)
    xh = Matrix{Float64}(undef, n, N); q = qmfvec(b.wt)
    check(ccall((:wx_isiwpd_f64, LIB), Cint,
                (Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Int64, Cint, Cint, Int64, Ptr{Float64}, Cint, Cint, Ptr{Cvoid}),
                b.W, b.status, xh, n, b.L, b.d, N, q, length(q), literal, C_NULL))
    return xh
end
function siwt_node(b::SIWTBatch, sig::Integer, j::Integer, i::Integer, t::Integer)
    n = size(b.W, 1); m = max(0, j - b.d); slot = t >> m; len = n >> j
    v = b.W[i*len+1:(i+1)*len, siwt_coloff(j, b.d) + slot + 1, sig]
    return WaveletsExt.SIWT.ShiftInvariantWaveletTransformNode{1,Int,Float64}(j, i, t,
        b.costs[siwt_nodeoff(j, b.d) + (slot << j) + i + 1, sig], v)
end

# ---- multi-GPU (one process per GPU; include/waveletsext_hip.h "Multi-GPU exchange") ----------------------------
# The launcher (MPI.jl, Distributed.jl) broadcasts the 128-byte id made on rank 0.  Transforms need no
# collective: each process runs the methods above on its contiguous shard `x[:, lo:hi]`.  `buf` arguments of the
# two exchange steps are device pointers (e.g. `pointer(::ROCArray)`), `count` elements per rank.
struct Comm; handle::Ptr{Cvoid}; nranks::Int; rank::Int; end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:wx_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end
function Comm(nranks::Integer, rank::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:wx_comm_init, LIB), Cint, (Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}), nranks, rank, id, h))
    return Comm(h[], nranks, rank)
end
Base.close(c::Comm) = check(ccall((:wx_comm_destroy, LIB), Cint, (Ptr{Cvoid},), c.handle))
# C1: reconstructed output shards -> full batch on every rank (recv holds nranks*count elements)
allgather_out!(recv::Ptr{Float64}, send::Ptr{Float64}, count::Integer, c::Comm, stream = C_NULL) =
    check(ccall((:wx_allgather_out_f64, LIB), Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                send, recv, count, c.handle, stream))
# C2: JBB moments [sum | sumsq] summed over ranks in place, then wx_jbb_costs_* / wx_treeselect_* on every rank
allreduce_moments!(buf::Ptr{Float64}, count::Integer, c::Comm, stream = C_NULL) =
    check(ccall((:wx_allreduce_moments_f64, LIB), Cint, (Ptr{Float64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                buf, count, c.handle, stream))
shutdown() = check(ccall((:wx_shutdown, LIB), Cint, ()))

end # module
