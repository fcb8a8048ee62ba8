// wx_bb.hip -- standard (per-signal) best basis, BB: SURVEY section 8(f) row 3.
//   tree_costs(X, ::BB)            bestbasis/bestbasis_tree.jl:210-258
//   coefcost(x, ::BBCost, nrm)     bestbasis/bestbasis_costs.jl:104-125 (Shannon / log-energy entropy)
//   bestbasis_treeselection        BestBasis.jl:59-110, delete_subtree! :128-140
//   bestbasistreeall(X, ::BB)      BestBasis.jl:253-262  (the batch loop: one tree per signal)
// Every signal is independent: costs are one workgroup per (node, signal), the tree selection one
// workgroup per signal with the signal's cost vector in LDS.
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include "wx_bbcost.h"

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

// norm of the root (first column / slice) of every signal: nrm[sig] = sqrt(sum x^2)
template <typename T>
__global__ __launch_bounds__(256) void k_bb_norms(const T *__restrict__ X, int64_t cnt, int64_t sig_stride,
                                                  T *__restrict__ nrm)
{
    __shared__ double red[256];
    const T *x = X + (int64_t)blockIdx.x * sig_stride;
    const double acc = wx_sumsq_strided<T>(x, cnt);
    const double tot = bb_block_sum(acc, red);
    if (threadIdx.x == 0) nrm[blockIdx.x] = (T)sqrt(tot);
}

// 1-D: X (n, k, batch).  Redundant: one workgroup per (node = column, signal).  Packet table: one workgroup
// per (level, signal) walks the level's column once -- nodes of >= 256 coefficients are reduced by the whole
// workgroup one after the other, smaller nodes by one thread each -- so the table is read exactly once and
// deep levels do not launch one workgroup per two-sample node.
// short signals (n <= 256), packet tables: the table is one contiguous array of (n, k, batch) -- thread g takes element g, the nodes of a level
// are groups of cnt = n >> depth consecutive lanes (segmented reduction by wavefront shuffles, one LDS step for nodes of 128 / 256), so a
// workgroup covers 256 / n (level, signal) rows at once instead of one row with most of its lanes idle (64-sample signals: 0.7 ms per 0.25 GiB
// table and launch -- 524 k workgroups -- against 0.1)
template <typename T>
__global__ __launch_bounds__(256) void k_bb_costs1d_short(const T *__restrict__ X, const T *__restrict__ nrm, int n, int k, int log2n, int cost_kind,
                                                          int64_t ncost, int64_t total, T *__restrict__ costs)
{
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = g < total;
    const int64_t row = (in ? g : total - 1) >> log2n;            // (signal, level) row; a workgroup never straddles rows longer than itself
    const int i = (int)(g & (n - 1));
    const int64_t sig = row / k;
    const int depth = (int)(row - sig * k);
    const int cnt = n >> depth;
    const T nr = nrm[sig];
    double v = 0.0;
    if (in && nr != (T)0) {
        const WxNorm<T> nrw(nr);
        v = bb_term<T>(X[g], nrw, cost_kind);
    }
    const int w0 = cnt < 64 ? cnt : 64;
    for (int w = w0 >> 1; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
    T *o = costs + sig * ncost + (((int64_t)1 << depth) - 1);
    if (cnt <= 64 && in && (lane & (cnt - 1)) == 0) o[i / cnt] = (T)v;
    // nodes of 128 / 256 coefficients: 2 / 4 wavefronts of this workgroup (the rows of a workgroup may be of different depths: every
    // wavefront passes the barrier, the ones with short nodes are done)
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const int per = cnt >> 6;                                   // wavefronts per node (0: nothing left to do)
    if (per >= 2 && lane == 0 && (wave % per) == 0 && in) {
        double t = 0.0;
        for (int q = 0; q < per; ++q) t += red[wave + q];
        o[i / cnt] = (T)t;
    }
}

// short signals, round 6: ONE WAVEFRONT per signal (four per workgroup) reads the signal's table row by row -- norm of the root, then the costs of
// every node: a level's row is NCH chunks of 64 coefficients, a node of cnt >= 64 coefficients is whole chunks (summed in the lane, then over the
// wavefront), shorter nodes are groups of cnt lanes (butterfly over cnt lanes).  No norm kernel, no per-element index division, the next row's
// loads fly under this row's logarithms.  (k_bb_norms + k_bb_costs1d_short: 0.94 ms per GiB table of 64-sample signals in five launches each;
// this kernel: profiles/r06_floor_misc.txt.)
template <typename T> __device__ __forceinline__ double bb_wave_sum(double v, int width)
{
    for (int w = width >> 1; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
    return v;
}
// ALL (64 ... 256 samples): every row of the table is loaded before the first logarithm (7 ... 36 values per lane) -- with one row in flight per
// wavefront the 64-sample kernel was bound by its seven dependent load latencies (0.69 ms per GiB table)
template <typename T, int NCH, bool ALL>
__global__ __launch_bounds__(256) void k_bb_costs1d_wave(const T *__restrict__ X, int k, int cost_kind, int64_t ncost, int64_t batch,
                                                         T *__restrict__ costs)
{
    constexpr int n = 64 * NCH;
    constexpr int KM = NCH == 1 ? 7 : (NCH == 2 ? 8 : (NCH == 4 ? 9 : 10));      // log2(n) + 1 rows at most
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sig = (int64_t)blockIdx.x * 4 + wave;
    if (sig >= batch) return;
    const T *x = X + sig * (int64_t)k * n;
    T *out = costs + sig * ncost;
    T rows[ALL ? KM : 2][NCH];
    if constexpr (ALL) {
#pragma unroll
        for (int d = 0; d < KM; ++d)
            if (d < k) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) rows[d][c] = x[(int64_t)d * n + 64 * c + lane];
            }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) rows[0][c] = x[64 * c + lane];
    }
    double a2 = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) { const double d = (double)rows[0][c]; a2 = fma(d, d, a2); }
    const T nr = (T)sqrt(bb_wave_sum<T>(a2, 64));
    if (nr == (T)0) {
        for (int64_t i = lane; i < ncost; i += 64) out[i] = (T)0.0;
        return;
    }
    const WxNorm<T> nrw(nr);
    auto level = [&](int depth, const T (&cur)[NCH]) {
        const int cnt = n >> depth;
        T *o = out + (((int64_t)1 << depth) - 1);
        double t[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) t[c] = bb_term<T>(cur[c], nrw, cost_kind);
        if (cnt >= 64) {
            const int per = cnt >> 6;                              // chunks per node
#pragma unroll
            for (int c0 = 0; c0 < NCH; ++c0) {
                if (c0 % per) continue;
                double sacc = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (c >= c0 && c < c0 + per) sacc += t[c];
                const double tot = bb_wave_sum<T>(sacc, 64);
                if (lane == 0) o[c0 / per] = (T)tot;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const double tot = bb_wave_sum<T>(t[c], cnt);
                if ((lane & (cnt - 1)) == 0) o[(64 * c + lane) / cnt] = (T)tot;
            }
        }
    };
    if constexpr (ALL) {
#pragma unroll
        for (int d = 0; d < KM; ++d)
            if (d < k) level(d, rows[d]);
    } else {
        for (int depth = 0; depth < k; ++depth) {
            if (depth + 1 < k) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) rows[1][c] = x[(int64_t)(depth + 1) * n + 64 * c + lane];
            }
            level(depth, rows[0]);
#pragma unroll
            for (int c = 0; c < NCH; ++c) rows[0][c] = rows[1][c];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_bb_costs1d(const T *__restrict__ X, const T *__restrict__ nrm, int n, int k,
                                                    int redundant, int cost_kind, int64_t ncost,
                                                    T *__restrict__ costs)
{
    __shared__ double red[256];
    const int64_t sig = blockIdx.y;
    const T nr = nrm[sig];
    const WxNorm<T> nrw(nr);
    T *out = costs + sig * ncost;
    if (redundant) {
        const int64_t idx = blockIdx.x;
        int depth = 0;
        { int64_t t = idx + 1; while (t >= 2) { t >>= 1; ++depth; } }
        const T *x = X + (sig * k + idx) * (int64_t)n;
        double acc = 0.0;
        if (nr != (T)0)
            for (int i = threadIdx.x; i < n; i += blockDim.x) acc += bb_term<T>(x[i], nrw, cost_kind);
        const double tot = bb_block_sum(acc, red);
        if (threadIdx.x == 0) out[idx] = (T)((T)(nr == (T)0 ? 0.0 : tot) / (T)((int64_t)1 << depth));
        return;
    }
    const int depth = blockIdx.x;                     // level = column
    const int cnt = n >> depth;                       // coefficients per node
    const int nodes = 1 << depth;
    const T *x = X + (sig * k + depth) * (int64_t)n;
    T *o = out + ((int64_t)1 << depth) - 1;
    if (cnt >= 256) {
        for (int node = 0; node < nodes; ++node) {
            double acc = 0.0;
            if (nr != (T)0) {
                // eight loads in flight per lane, then the (long) cost terms
                const T *xn = x + (int64_t)node * cnt;
                int i = threadIdx.x;
                for (; i + 7 * 256 < cnt; i += 8 * 256) {
                    T v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = xn[i + u * 256];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += bb_term<T>(v[u], nrw, cost_kind);
                }
                for (; i < cnt; i += 256) acc += bb_term<T>(xn[i], nrw, cost_kind);
            }
            const double tot = bb_block_sum(acc, red);
            if (threadIdx.x == 0) o[node] = (T)(nr == (T)0 ? 0.0 : tot);
        }
    } else if (n >= 256) {
        // coalesced chunks of 256 consecutive coefficients = 256/cnt whole nodes: segmented reduction with
        // wavefront shuffles (groups of cnt <= 64 lanes), one more LDS step for 128-sample nodes
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        T nextv = x[threadIdx.x];                              // the next chunk's load flies under this chunk's terms
        for (int base = 0; base < n; base += 256) {
            const T cur = nextv;
            if (base + 256 < n) nextv = x[base + 256 + threadIdx.x];
            double v = nr != (T)0 ? bb_term<T>(cur, nrw, cost_kind) : 0.0;
            const int w0 = cnt < 64 ? cnt : 64;
            for (int w = w0 >> 1; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
            if (cnt <= 64) {
                if ((lane & (cnt - 1)) == 0) o[(base + (int)threadIdx.x) / cnt] = (T)v;
            } else {                                           // cnt == 128: two wavefronts per node
                if (lane == 0) red[wave] = v;
                __syncthreads();
                if (threadIdx.x < 2) o[base / cnt + threadIdx.x] = (T)(red[2 * threadIdx.x] + red[2 * threadIdx.x + 1]);
                __syncthreads();
            }
        }
    } else {
        for (int node = threadIdx.x; node < nodes; node += blockDim.x) {
            double acc = 0.0;
            if (nr != (T)0)
                for (int i = 0; i < cnt; ++i) acc += bb_term<T>(x[(int64_t)node * cnt + i], nrw, cost_kind);
            o[node] = (T)(nr == (T)0 ? 0.0 : acc);
        }
    }
}

// 2-D: X (m, n, k, batch).  Redundant: slice idx with the signal's norm, / 4^depth, one workgroup per
// (node, signal).  Packet table: one workgroup per (level, signal); every block is normalised by its own
// norm (bestbasis_tree.jl:252 passes no nrm); big blocks by the workgroup, small ones by one thread each.
template <typename T>
__global__ __launch_bounds__(256) void k_bb_costs2d(const T *__restrict__ X, const T *__restrict__ nrm, int m, int n,
                                                    int k, int redundant, int cost_kind, int64_t ncost,
                                                    T *__restrict__ costs)
{
    __shared__ double red[256];
    const int64_t sig = blockIdx.y;
    T *out = costs + sig * ncost;
    if (redundant) {
        const int64_t idx = blockIdx.x;
        int depth = 0;
        { int64_t t = 3 * (idx + 1) - 2; while (t >= 4) { t >>= 2; ++depth; } }
        const T *x = X + (sig * k + idx) * (int64_t)m * n;
        const T nv = nrm[sig];
        const WxNorm<T> nvw(nv);
        const int cnt = m * n;
        double acc = 0.0;
        if (nv != (T)0)
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) acc += bb_term<T>(x[i], nvw, cost_kind);
        const double tot = bb_block_sum(acc, red);
        if (threadIdx.x == 0) out[idx] = (T)((T)(nv == (T)0 ? 0.0 : tot) / (T)((int64_t)1 << (2 * depth)));
        return;
    }
    const int depth = blockIdx.x;
    const int nr_ = m >> depth, ncl = n >> depth, cnt = nr_ * ncl;
    const int nodes = 1 << (2 * depth);
    const T *x = X + (sig * k + depth) * (int64_t)m * n;
    T *o = out + ((((int64_t)1 << (2 * depth)) - 1) / 3);
    // node `mort` (morton code within the level: row bit above column bit at every depth) -> block origin
    auto origin = [&](int mort, int &r0, int &c0) {
        int jr = 0, jc = 0;
        for (int t = 0; t < depth; ++t) { jr |= ((mort >> (2 * t + 1)) & 1) << t; jc |= ((mort >> (2 * t)) & 1) << t; }
        r0 = jr * nr_; c0 = jc * ncl;
    };
    if (cnt >= 256) {
        for (int node = 0; node < nodes; ++node) {
            int r0, c0;
            origin(node, r0, c0);
            double a2 = 0.0;
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
                const double v = (double)x[(int64_t)(c0 + i / nr_) * m + r0 + i % nr_];
                a2 = fma(v, v, a2);
            }
            const T nv = (T)sqrt(bb_block_sum(a2, red));
            const WxNorm<T> nvw(nv);
            double acc = 0.0;
            if (nv != (T)0)
                for (int i = threadIdx.x; i < cnt; i += blockDim.x)
                    acc += bb_term<T>(x[(int64_t)(c0 + i / nr_) * m + r0 + i % nr_], nvw, cost_kind);
            const double tot = bb_block_sum(acc, red);
            if (threadIdx.x == 0) o[node] = (T)(nv == (T)0 ? 0.0 : tot);
        }
    } else {
        for (int node = threadIdx.x; node < nodes; node += blockDim.x) {
            int r0, c0;
            origin(node, r0, c0);
            double a2 = 0.0;
            for (int c = 0; c < ncl; ++c)
                for (int r = 0; r < nr_; ++r) { const double v = (double)x[(int64_t)(c0 + c) * m + r0 + r]; a2 = fma(v, v, a2); }
            const T nv = (T)sqrt(a2);
            const WxNorm<T> nvw(nv);
            double acc = 0.0;
            if (nv != (T)0)
                for (int c = 0; c < ncl; ++c)
                    for (int r = 0; r < nr_; ++r) acc += bb_term<T>(x[(int64_t)(c0 + c) * m + r0 + r], nvw, cost_kind);
            o[node] = (T)(nv == (T)0 ? 0.0 : acc);
        }
    }
}

// Bottom-up selection, one workgroup per signal.  The sequential reference (i = ntree..1: keep the
// children's sum when it beats the parent, else delete_subtree!(i)) is level-synchronous: a node's
// decision needs only its children's final costs, and deleting a subtree clears exactly the nodes that
// have a pruned ancestor-or-self, so   tree[i] = full[i] && !pruned[i] && tree[parent(i)].
// costs (T, mutated like the reference) and the flags live in LDS.
// GM: cost vectors that do not fit a CU's LDS (signals of more than about 8000 samples' worth of nodes) are updated where they are, in
// global memory, with the flags in a scratch row (the reference has no limit: BestBasis.jl:253-262)
template <typename T, int ARITY, bool GM>
__global__ __launch_bounds__(256) void k_bb_treeselect(T *__restrict__ costs, int64_t ncost, int L, int64_t ntree,
                                                       int type_max, uint8_t *__restrict__ trees, uint8_t *__restrict__ gflags)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    const int64_t sig = blockIdx.x;
    T *gc = costs + sig * ncost;
    T *c = GM ? gc : reinterpret_cast<T *>(wx_smem);
    uint8_t *flag = GM ? gflags + sig * ncost : reinterpret_cast<uint8_t *>(reinterpret_cast<T *>(wx_smem) + ncost);   // 1 = pruned, later the tree bit
    if (!GM) {
        // eight loads of a lane in flight together
        int64_t i = threadIdx.x;
        for (; i + 7 * 256 < ncost; i += 8 * 256) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gc[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[i + u * 256] = v[u];
        }
        for (; i < ncost; i += 256) c[i] = gc[i];
    }
    __syncthreads();
    // first 1-based index of depth d: binary 2^d, quad (4^d - 1)/3 + 1
    auto first = [](int d) -> int64_t { return ARITY == 2 ? ((int64_t)1 << d) : ((((int64_t)1 << (2 * d)) - 1) / 3 + 1); };
    for (int d = L - 1; d >= 0; --d) {
        const int64_t lo = first(d), hi = first(d + 1);
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const T pc = c[i - 1];
            T cc;
            if (ARITY == 2) cc = (T)(c[2 * i - 1] + c[2 * i]);
            else cc = (T)((T)((T)(c[4 * i - 3] + c[4 * i - 2]) + c[4 * i - 1]) + c[4 * i]);
            const bool better = type_max ? (cc > pc) : (cc < pc);
            if (better) c[i - 1] = cc;
            flag[i - 1] = better ? 0 : 1;
        }
        __syncthreads();
    }
    // top down, in place: flag[] of shallower depths already holds the tree bit
    for (int d = 0; d < L; ++d) {
        const int64_t lo = first(d), hi = first(d + 1);
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            bool keep = !flag[i - 1];
            if (i > 1) {
                const int64_t par = ARITY == 2 ? (i >> 1) : ((i + 2) >> 2);
                keep = keep && flag[par - 1];
            }
            flag[i - 1] = keep ? 1 : 0;
        }
        __syncthreads();
    }
    uint8_t *out = trees + sig * ntree;
    const int64_t nfull = first(L) - 1;
    for (int64_t i = threadIdx.x; i < ntree; i += blockDim.x) out[i] = i < nfull ? flag[i] : 0;
    if (!GM)
        for (int64_t i = threadIdx.x; i < ncost; i += blockDim.x) gc[i] = c[i];
}

// binary trees of at most 511 nodes (signals of up to 256 samples): ONE WAVEFRONT per signal, four per workgroup, the same two sweeps with
// wave-level ordering instead of workgroup barriers (a 256-thread workgroup per 64-sample signal -- 127 costs, 12 barriers -- took 0.58 ms per
// GiB table; this kernel: profiles/r06_floor_misc.txt)
template <typename T>
__global__ __launch_bounds__(256) void k_bb_treeselect_w(T *__restrict__ costs, int64_t ncost, int L, int64_t ntree, int type_max, int64_t batch,
                                                         uint8_t *__restrict__ trees)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sig = (int64_t)blockIdx.x * 4 + wave;
    if (sig >= batch) return;
    const size_t per = ((size_t)ncost * (sizeof(T) + 1) + 15) & ~(size_t)15;
    T *c = reinterpret_cast<T *>(wx_smem + wave * per);
    uint8_t *flag = reinterpret_cast<uint8_t *>(c + ncost);                     // 1 = pruned, later the tree bit
    T *gc = costs + sig * ncost;
    auto wsync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int64_t i = lane; i < ncost; i += 64) c[i] = gc[i];
    wsync();
    for (int d = L - 1; d >= 0; --d) {
        const int64_t lo = (int64_t)1 << d, hi = (int64_t)2 << d;
        for (int64_t i = lo + lane; i < hi; i += 64) {
            const T pc = c[i - 1];
            const T cc = (T)(c[2 * i - 1] + c[2 * i]);
            const bool better = type_max ? (cc > pc) : (cc < pc);
            if (better) c[i - 1] = cc;
            flag[i - 1] = better ? 0 : 1;
        }
        wsync();
    }
    for (int d = 0; d < L; ++d) {
        const int64_t lo = (int64_t)1 << d, hi = (int64_t)2 << d;
        for (int64_t i = lo + lane; i < hi; i += 64) {
            bool keep = !flag[i - 1];
            if (i > 1) keep = keep && flag[(i >> 1) - 1];
            flag[i - 1] = keep ? 1 : 0;
        }
        wsync();
    }
    uint8_t *out = trees + sig * ntree;
    const int64_t nfull = ((int64_t)1 << L) - 1;
    for (int64_t i = lane; i < ntree; i += 64) out[i] = i < nfull ? flag[i] : 0;
    for (int64_t i = lane; i < ncost; i += 64) gc[i] = c[i];
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

template <typename T>
int api_bb_costs(const T *X, T *costs, int64_t m, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind,
                 bool two_d, void *stream)
{
    WX_REQUIRE(m >= 1 && n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(cost_kind == 0 || cost_kind == 1, WX_EARG, "cost_kind must be 0 (ShannonEntropyCost) or 1 (LogEnergyEntropyCost)");
    const int64_t sigsz = two_d ? m * n : m;
    if (!redundant) {
        WX_REQUIRE(k <= (two_d ? 15 : 30) && k - 1 <= wx_maxtransformlevels(two_d ? (m < n ? m : n) : m), WX_EASSERT,
                   "more packet levels than the signal admits (nodelength)");
    }
    const int64_t ncost = redundant ? k : (two_d ? ((((int64_t)1 << (2 * k)) - 1) / 3) : (((int64_t)1 << k) - 1));
    WX_REQUIRE(sigsz < ((int64_t)1 << 31) && ncost < ((int64_t)1 << 31) && batch <= 65535 * (int64_t)65535, WX_EUNSUPPORTED,
               "signal or tree too large");
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * sigsz * k * batch);
    T *dc = (T *)io.out(costs, sizeof(T) * ncost * batch);
    if (!dX || !dc) return io.finish(WX_EHIP);
    T *dn = (T *)scr.alloc(sizeof(T) * batch);
    if (!dn) return io.finish(WX_EHIP);
    static const bool wave_off = wx_getenv("WX_BB_WAVE") && atoi(wx_getenv("WX_BB_WAVE")) == 0;
    if (!two_d && !redundant && !wave_off && (m == 64 || m == 128 || m == 256 || m == 512) && batch <= 0x7ffffff0) {
        const dim3 g((unsigned)((batch + 3) / 4));
        switch (m) {
        case 64: hipLaunchKernelGGL((k_bb_costs1d_wave<T, 1, true>), g, dim3(256), 0, st, dX, (int)k, cost_kind, ncost, batch, dc); break;
        case 128: hipLaunchKernelGGL((k_bb_costs1d_wave<T, 2, true>), g, dim3(256), 0, st, dX, (int)k, cost_kind, ncost, batch, dc); break;
        case 256: hipLaunchKernelGGL((k_bb_costs1d_wave<T, 4, true>), g, dim3(256), 0, st, dX, (int)k, cost_kind, ncost, batch, dc); break;
        default: hipLaunchKernelGGL((k_bb_costs1d_wave<T, 8, false>), g, dim3(256), 0, st, dX, (int)k, cost_kind, ncost, batch, dc); break;
        }
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "BB cost kernels failed to launch"));
        return io.finish(WX_OK);
    }
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {                   // gridDim.y limit
        const int64_t bc = batch - b0 < 65535 ? batch - b0 : 65535;
        const T *xs = dX + b0 * sigsz * k;
        if (!two_d || redundant)
            hipLaunchKernelGGL(k_bb_norms<T>, dim3((unsigned)bc), dim3(256), 0, st, xs, sigsz, sigsz * k, dn + b0);
        if (!two_d && !redundant && m <= 256 && m >= 2 && !(m & (m - 1)) && !wx_getenv("WX_BB_SHORT_OFF")) {
            int lg = 0;
            while (((int64_t)1 << lg) < m) ++lg;
            const int64_t total = bc * k * m;
            hipLaunchKernelGGL(k_bb_costs1d_short<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, xs, (const T *)(dn + b0), (int)m, (int)k, lg,
                               cost_kind, ncost, total, dc + b0 * ncost);
        } else if (!two_d)
            hipLaunchKernelGGL(k_bb_costs1d<T>, dim3((unsigned)(redundant ? ncost : k), (unsigned)bc), dim3(256), 0, st, xs, (const T *)(dn + b0),
                               (int)m, (int)k, redundant, cost_kind, ncost, dc + b0 * ncost);
        else
            hipLaunchKernelGGL(k_bb_costs2d<T>, dim3((unsigned)(redundant ? ncost : k), (unsigned)bc), dim3(256), 0, st, xs, (const T *)(dn + b0),
                               (int)m, (int)n, (int)k, redundant, cost_kind, ncost, dc + b0 * ncost);
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "BB cost kernels failed to launch"));
    return io.finish(WX_OK);
}

template <typename T>
int api_treeselect_batch(T *costs, int64_t ncost, int64_t m, int64_t n, int type_max, int64_t batch, uint8_t *trees,
                         void *stream)
{
    const bool two_d = n > 0;
    WX_REQUIRE(costs && trees, WX_EARG, "NULL argument");
    WX_REQUIRE(type_max == 0 || type_max == 1, WX_EARG, "Unsupported type (BestBasis.jl:64,91)");
    WX_REQUIRE(m >= 1 && ncost >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    int L;
    int64_t ntree;
    if (!two_d) {
        const int64_t tl = ((int64_t)1 << wx_maxtransformlevels(2 * m)) - 1;
        WX_REQUIRE(ncost <= tl, WX_EASSERT, "@assert k <= gettreelength(2*n) (BestBasis.jl:63)");
        L = wx_getdepth_binary(ncost);
        WX_REQUIRE(wx_isdyadic(m) && L <= wx_maxtransformlevels(m), WX_EASSERT, "maketree(n, L, :full)");
        ntree = m - 1;
        WX_REQUIRE(ncost >= ((int64_t)1 << (L + 1)) - 1 || L == 0, WX_EBOUNDS, "costs do not cover the children of depth L-1");
    } else {
        WX_REQUIRE(ncost <= wx_gettreelength2d(2 * m, 2 * n), WX_EASSERT, "@assert k <= gettreelength(2*n,2*m) (BestBasis.jl:90)");
        L = wx_getdepth_quad(ncost);
        WX_REQUIRE(L <= wx_maxtransformlevels(m < n ? m : n), WX_EASSERT, "maketree(n, m, L, :full)");
        ntree = wx_gettreelength2d(m, n);
        WX_REQUIRE(ncost >= ((((int64_t)1 << (2 * (L + 1))) - 1) / 3) || L == 0, WX_EBOUNDS, "costs do not cover the children of depth L-1");
    }
    const size_t lds = (size_t)ncost * sizeof(T) + (size_t)ncost + 16;
    const bool gm = lds > 150 * 1024;
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0 || ntree == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    T *dc = (T *)io.in(costs, sizeof(T) * ncost * batch);
    for (auto &it : io.items) if (it.user == costs) it.copy_out = true;      // costs are mutated like the reference
    uint8_t *dt = (uint8_t *)io.out(trees, (size_t)ntree * batch);
    if (!dc || !dt) return io.finish(WX_EHIP);
    if (gm) {
        WxScratch scr(st);
        // flags of at most 1 GiB of signals at a time
        int64_t per = ((int64_t)1 << 30) / ncost;
        if (per < 1) per = 1;
        if (per > batch) per = batch;
        uint8_t *gflags = (uint8_t *)scr.alloc((size_t)per * ncost);
        if (!gflags) return io.finish(WX_EHIP);
        auto kg = two_d ? k_bb_treeselect<T, 4, true> : k_bb_treeselect<T, 2, true>;
        for (int64_t b0 = 0; b0 < batch; b0 += per) {
            const int64_t nb = batch - b0 < per ? batch - b0 : per;
            hipLaunchKernelGGL(kg, dim3((unsigned)nb), dim3(256), 0, st, dc + b0 * ncost, ncost, L, ntree, type_max, dt + b0 * ntree, gflags);
        }
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "tree selection kernel failed to launch"));
        return io.finish(WX_OK);
    }
    static const bool wave_off = wx_getenv("WX_BB_WAVE") && atoi(wx_getenv("WX_BB_WAVE")) == 0;
    if (!two_d && !wave_off && ncost <= 511 && batch <= 0x7ffffff0) {      // up to 256 samples (512: 0.125 against 0.109 ms)
        const size_t per = ((size_t)ncost * (sizeof(T) + 1) + 15) & ~(size_t)15;
        hipLaunchKernelGGL(k_bb_treeselect_w<T>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 4 * per, st, dc, ncost, L, ntree, type_max, batch, dt);
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "tree selection kernel failed to launch"));
        return io.finish(WX_OK);
    }
    auto kern = two_d ? k_bb_treeselect<T, 4, false> : k_bb_treeselect<T, 2, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(256), lds, st, dc, ncost, L, ntree, type_max, dt, (uint8_t *)nullptr);
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "tree selection kernel failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_bb_costs_f64(const double *X, double *costs, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind, void *stream)
{ return api_bb_costs<double>(X, costs, n, 1, k, batch, redundant, cost_kind, false, stream); }
int wx_bb_costs_f32(const float *X, float *costs, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind, void *stream)
{ return api_bb_costs<float>(X, costs, n, 1, k, batch, redundant, cost_kind, false, stream); }
int wx_bb_costs2d_f64(const double *X, double *costs, int64_t m, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind,
                      void *stream)
{ return api_bb_costs<double>(X, costs, m, n, k, batch, redundant, cost_kind, true, stream); }
int wx_bb_costs2d_f32(const float *X, float *costs, int64_t m, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind,
                      void *stream)
{ return api_bb_costs<float>(X, costs, m, n, k, batch, redundant, cost_kind, true, stream); }
int wx_treeselect_batch_f64(double *costs, int64_t ncost, int64_t m, int64_t n, int type_max, int64_t batch, uint8_t *trees, void *stream)
{ return api_treeselect_batch<double>(costs, ncost, m, n, type_max, batch, trees, stream); }
int wx_treeselect_batch_f32(float *costs, int64_t ncost, int64_t m, int64_t n, int type_max, int64_t batch, uint8_t *trees, void *stream)
{ return api_treeselect_batch<float>(costs, ncost, m, n, type_max, batch, trees, stream); }
}
