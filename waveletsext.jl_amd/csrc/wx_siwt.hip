// wx_siwt.hip -- shift-invariant wavelet packet decomposition (Cohen, Raz & Malah), SURVEY section 8(f) row 4.
//   siwpd / siwpd_subtree!                      SIWT.jl:57-137
//   sidwt_step! / isidwt_step!                  siwt/siwt_one_level.jl:71-98, 154-185
//   node cost (ShannonEntropyCost, signal norm) siwt/siwt_utls.jl:118-126, bestbasis/bestbasis_costs.jl:104-124
//   bestbasistree! / bestbasis_treeselection!   siwt/siwt_bestbasis.jl:28-102
//   isiwpd / isiwpd_subtree!                    SIWT.jl:166-229
//
// The reference keeps a Dict of node objects per signal and recurses.  Here a batch of signals shares one flat
// table W (n, NS, batch): a node (Depth j, IndexAtDepth i, TransformShift t) is the i-th block of n >> j samples
// of the column that holds "depth j, shift t", i.e. every column is one level of the packet decomposition of the
// signal rotated by t.  The shifts that exist at depth j are the multiples of 2^max(0, j-d) below 2^j (d = depth
// of the shifted transforms, SIWT.jl:119-131 in closed form), so depth j owns 2^min(j,d) columns and a level is
// one launch over all of its columns and all signals; costs, the three-way (node / children / shifted children)
// selection and the inverse are level-synchronous launches as well.  Built with -ffp-contract=off and the
// reference's tap order: the Float64 coefficients are bit-identical to the scalar restatement in oracle/.
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include "wx_bbcost.h"

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

struct WxSiwtGeom {
    int L, d;
    int64_t coloff[34];       // first column of depth j; coloff[L+1] = NS
    int64_t nodeoff[34];      // first node (cost / status index) of depth j; nodeoff[L+1] = NN
};

WxSiwtGeom siwt_geom(int L, int d)
{
    WxSiwtGeom g;
    g.L = L; g.d = d;
    int64_t c = 0, nn = 0;
    for (int j = 0; j <= L + 1; ++j) {
        g.coloff[j] = c; g.nodeoff[j] = nn;
        const int64_t ns = (int64_t)1 << (j < d ? j : d);
        c += ns; nn += ns << j;
    }
    return g;
}

// one multiply-add of the reference's `w += g * v`: filters are Float64, the product and the sum are formed in
// Float64 and the store rounds to T; no contraction in this translation unit
template <typename T> __device__ __forceinline__ T siwt_mac(T acc, double q, T v) { return (T)((double)acc + q * (double)v); }

// ---- forward: depth j -> j + 1 -------------------------------------------------------------------------
// item = (child slot, node, i0): a = sum_t q[t] v[(2 i0 - s + t) mod np], d = sum_t (-1)^t q[t] v[(2 i0 + 1 - s - t) mod np]
// FUSE: the Shannon cost of the two child nodes is reduced in the same launch (children of at most 256 samples,
// dyadic n: a child is a run of h2 consecutive lanes -- wavefront shuffles, then one LDS step across the
// wavefronts of a 128 / 256-sample child), so the table is not read a second time for the costs of these depths.
// FT: filter length known at compile time (taps in scalar registers, all loads of a lane in flight), 0 = any length
template <typename T, bool FUSE, int FT>
__global__ __launch_bounds__(256) void k_siwt_fwd_level(T *__restrict__ W, int n, int64_t NS, int j, int d,
                                                        int64_t col_j, int64_t col_j1, int64_t items, WxFilt filt,
                                                        const T *__restrict__ nrm, T *__restrict__ costs, int64_t NN,
                                                        int64_t node_j1)
{
    __shared__ double red[2][4];
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = e < items;
    if (!FUSE && !valid) return;
    if (!valid) e = items - 1;                                      // keeps the lane in the reductions; adds nothing
    const int half = n >> 1;
    const int64_t slot1 = e / half;
    const int r = (int)(e - slot1 * half);
    const int np = n >> j, h2 = np >> 1;
    const int node = r / h2, i0 = r - node * h2;
    const int m1 = j + 1 > d ? j + 1 - d : 0, m0 = j > d ? j - d : 0;
    const int64_t shift1 = slot1 << m1;
    const int s = (int)((shift1 >> j) & 1);
    const int64_t slot0 = (shift1 & (((int64_t)1 << j) - 1)) >> m0;
    T *sig = W + (int64_t)blockIdx.y * NS * n;
    const T *v = sig + (col_j + slot0) * n + (int64_t)node * np;
    T *o = sig + (col_j1 + slot1) * n + (int64_t)node * np;
    const int F = FT ? FT : filt.F;
    int k1 = 2 * i0 - s; if (k1 < 0) k1 += np;
    int k2 = 2 * i0 + 1 - s;
    T a, dd;
    if (FT) {
        T xa[FT ? FT : 1], xd[FT ? FT : 1];
#pragma unroll
        for (int t = 0; t < FT; ++t) {
            if (t) { if (++k1 == np) k1 = 0; if (--k2 < 0) k2 = np - 1; }
            xa[t] = v[k1]; xd[t] = v[k2];
        }
        a = (T)(filt.q[0] * (double)xa[0]);
        dd = (T)(filt.q[0] * (double)xd[0]);
#pragma unroll
        for (int t = 1; t < FT; ++t) {
            a = siwt_mac<T>(a, filt.q[t], xa[t]);
            dd = siwt_mac<T>(dd, (t & 1) ? -filt.q[t] : filt.q[t], xd[t]);
        }
    } else {
    a = (T)(filt.q[0] * (double)v[k1]);
    dd = (T)(filt.q[0] * (double)v[k2]);
    // taps in blocks of eight: the sixteen loads of a block are in flight together (a tap-by-tap loop waits for
    // every pair of loads: measured 0.27 ms per level against 0.11 ms of HBM time); the sums keep their order
    for (int t0 = 1; t0 < F; t0 += 8) {
        T xa[8], xd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u < F) {
                if (++k1 == np) k1 = 0;
                if (--k2 < 0) k2 = np - 1;
                xa[u] = v[k1]; xd[u] = v[k2];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u < F) {
                const int t = t0 + u;
                a = siwt_mac<T>(a, filt.q[t], xa[u]);
                dd = siwt_mac<T>(dd, (t & 1) ? -filt.q[t] : filt.q[t], xd[u]);
            }
        }
    }
    }
    if (valid) {
        o[i0] = a;
        o[h2 + i0] = dd;
    }
    if (FUSE) {
        const T nr = nrm[blockIdx.y];
        const WxNorm<T> nrw(nr);
        double ca = (valid && nr != (T)0) ? bb_term<T>(a, nrw, 0) : 0.0;
        double cd = (valid && nr != (T)0) ? bb_term<T>(dd, nrw, 0) : 0.0;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int w0 = h2 < 64 ? h2 : 64;
        for (int w = w0 >> 1; w > 0; w >>= 1) { ca += __shfl_xor(ca, w, 64); cd += __shfl_xor(cd, w, 64); }
        T *c = costs + (int64_t)blockIdx.y * NN + node_j1 + (slot1 << (j + 1)) + 2 * node;
        if (h2 <= 64) {
            if (valid && (lane & (h2 - 1)) == 0) { c[0] = (T)ca; c[1] = (T)cd; }
        } else {                                                   // 128 or 256 samples: 2 or 4 wavefronts per child
            if (lane == 0) { red[0][wave] = ca; red[1][wave] = cd; }
            __syncthreads();
            const int wpn = h2 >> 6;
            if (valid && lane == 0 && (wave & (wpn - 1)) == 0) {
                double sa = 0.0, sd = 0.0;
                for (int u = 0; u < wpn; ++u) { sa += red[0][wave + u]; sd += red[1][wave + u]; }
                c[0] = (T)sa; c[1] = (T)sd;
            }
        }
    }
}

// ---- inverse: children of depth j + 1 -> the nodes of depth j that kept children ------------------------
// status: 0 not in the tree, 1 leaf, 2 non-shifted children, 3 shifted children.
// A valid tree keeps at most one shift per (depth, IndexAtDepth) -- the path from the root fixes it -- so the
// nodes of the tree are found through a per-signal map (depth, index) -> slot * 4 + status built in one pass over
// the status bytes; every level then runs n / 2 lanes per signal instead of one lane per pair of every column.
__global__ __launch_bounds__(256) void k_siwt_build_map(const uint8_t *__restrict__ status, int64_t NN, WxSiwtGeom g,
                                                        int *__restrict__ map, int mapn)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= NN) return;
    const uint8_t stt = status[(int64_t)blockIdx.y * NN + e];
    if (!stt) return;
    int j = 0;
    while (j < g.L && e >= g.nodeoff[j + 1]) ++j;
    const int64_t rel = e - g.nodeoff[j];
    const int slot = (int)(rel >> j), idx = (int)(rel & (((int64_t)1 << j) - 1));
    atomicMax(&map[(int64_t)blockIdx.y * mapn + ((1 << j) - 1) + idx], slot * 4 + stt);   // (max: defined even for an invalid tree)
}

template <typename T, int FT>
__global__ __launch_bounds__(256) void k_siwt_inv_level(T *__restrict__ W, const int *__restrict__ map, int mapn, int n,
                                                        int64_t NS, int j, int d, int64_t col_j, int64_t col_j1,
                                                        WxFilt filt, int literal)
{
    const int r = (int)(blockIdx.x * 256 + threadIdx.x);
    if (r >= (n >> 1)) return;
    const int np = n >> j, h2 = np >> 1;
    const int node = r / h2, k = r - node * h2;
    const int mv = map[(int64_t)blockIdx.y * mapn + ((1 << j) - 1) + node];
    if (mv < 0 || (mv & 3) < 2) return;
    const int64_t slot0 = mv >> 2;
    const int shifted = (mv & 3) == 3;
    // literal: the flag as siwt_one_level.jl:126 spells it (true for the non-shifted children), see DESIGN.md 4.13
    const int s = literal ? !shifted : shifted;
    const int m1 = j + 1 > d ? j + 1 - d : 0, m0 = j > d ? j - d : 0;
    const int64_t slot1 = ((slot0 << m0) + (shifted ? ((int64_t)1 << j) : 0)) >> m1;
    T *sig = W + (int64_t)blockIdx.y * NS * n;
    const T *a = sig + (col_j1 + slot1) * n + (int64_t)node * np;
    const T *dd = a + h2;
    T *v = sig + (col_j + slot0) * n + (int64_t)node * np;
    const int F = FT ? FT : filt.F;
    // isidwt_step!: v[l] = g*w1 + h*w2, then v[l] += (g*w1 + h*w2) per further tap pair
    int ka = k, kd = k;
    T ev, od;
    if (FT) {
        constexpr int HT = FT ? FT / 2 : 1;
        T xa[HT], xd[HT];
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            if (m) { if (--ka < 0) ka = h2 - 1; if (++kd == h2) kd = 0; }
            xa[m] = a[ka]; xd[m] = dd[kd];
        }
        ev = (T)(filt.q[0] * (double)xa[0] + (-filt.q[1]) * (double)xd[0]);
        od = (T)(filt.q[1] * (double)xa[0] + filt.q[0] * (double)xd[0]);
#pragma unroll
        for (int m = 1; m < HT; ++m) {
            ev = (T)((double)ev + (filt.q[2 * m] * (double)xa[m] + (-filt.q[2 * m + 1]) * (double)xd[m]));
            od = (T)((double)od + (filt.q[2 * m + 1] * (double)xa[m] + filt.q[2 * m] * (double)xd[m]));
        }
    } else {
    ev = (T)(filt.q[0] * (double)a[ka] + (-filt.q[1]) * (double)dd[kd]);
    od = (T)(filt.q[1] * (double)a[ka] + filt.q[0] * (double)dd[kd]);
    for (int m0 = 1; 2 * m0 < F; m0 += 8) {
        T xa[8], xd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (2 * (m0 + u) < F) {
                if (--ka < 0) ka = h2 - 1;
                if (++kd == h2) kd = 0;
                xa[u] = a[ka]; xd[u] = dd[kd];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (2 * (m0 + u) < F) {
                const int m = m0 + u;
                ev = (T)((double)ev + (filt.q[2 * m] * (double)xa[u] + (-filt.q[2 * m + 1]) * (double)xd[u]));
                od = (T)((double)od + (filt.q[2 * m + 1] * (double)xa[u] + filt.q[2 * m] * (double)xd[u]));
            }
        }
    }
    }
    int p0 = 2 * k - s; if (p0 < 0) p0 += np;
    v[p0] = ev;
    v[2 * k + 1 - s] = od;
}

// ---- node costs ------------------------------------------------------------------------------------------
// one workgroup per (column, signal); the column's nodes are reduced like the packet levels of k_bb_costs1d
template <typename T>
__global__ __launch_bounds__(256) void k_siwt_costs(const T *__restrict__ W, const T *__restrict__ nrm, int n, int64_t NS,
                                                    int64_t NN, WxSiwtGeom g, T *__restrict__ costs)
{
    __shared__ double red[256];
    const int64_t col = blockIdx.x, sig = blockIdx.y;
    int j = 0;
    while (j < g.L && col >= g.coloff[j + 1]) ++j;
    const int64_t slot = col - g.coloff[j];
    const T nr = nrm[sig];
    const WxNorm<T> nrw(nr);
    const int cnt = n >> j, nodes = 1 << j;
    const T *x = W + (sig * NS + col) * (int64_t)n;
    T *o = costs + sig * NN + g.nodeoff[j] + (slot << j);
    if (cnt >= 256) {
        for (int node = 0; node < nodes; ++node) {
            double acc = 0.0;
            if (nr != (T)0)
                for (int i = threadIdx.x; i < cnt; i += 256) acc += bb_term<T>(x[(int64_t)node * cnt + i], nrw, 0);
            const double tot = bb_block_sum(acc, red);
            if (threadIdx.x == 0) o[node] = (T)(nr == (T)0 ? 0.0 : tot);
        }
    } else if (n >= 256 && (cnt & (cnt - 1)) == 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int base = 0; base < n; base += 256) {
            const bool in = base + (int)threadIdx.x < n;
            double v = (in && nr != (T)0) ? bb_term<T>(x[base + threadIdx.x], nrw, 0) : 0.0;
            const int w0 = cnt < 64 ? cnt : 64;
            for (int w = w0 >> 1; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
            if (cnt <= 64) {
                if (in && (lane & (cnt - 1)) == 0) o[(base + (int)threadIdx.x) / cnt] = (T)v;
            } else {                                           // cnt == 128: two wavefronts per node
                if (lane == 0) red[wave] = v;
                __syncthreads();
                if (threadIdx.x < 2 && base + (int)threadIdx.x * 128 < n)
                    o[base / cnt + threadIdx.x] = (T)(red[2 * threadIdx.x] + red[2 * threadIdx.x + 1]);
                __syncthreads();
            }
        }
    } else {
        for (int node = threadIdx.x; node < nodes; node += 256) {
            double acc = 0.0;
            if (nr != (T)0)
                for (int i = 0; i < cnt; ++i) acc += bb_term<T>(x[(int64_t)node * cnt + i], nrw, 0);
            o[node] = (T)(nr == (T)0 ? 0.0 : acc);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_siwt_norms(const T *__restrict__ W, int n, int64_t sig_stride, T *__restrict__ nrm)
{
    __shared__ double red[256];
    const T *x = W + (int64_t)blockIdx.x * sig_stride;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double v = (double)x[i]; acc += v * v; }
    const double tot = bb_block_sum(acc, red);
    if (threadIdx.x == 0) nrm[blockIdx.x] = (T)sqrt(tot);
}

// ---- best basis: bottom-up choice, then top-down membership -------------------------------------------------
// choice[node]: 0 keep the node, 1 non-shifted children, 2 shifted children (siwt_bestbasis.jl:81-99: the node
// wins only when strictly cheaper than both pairs; the non-shifted pair wins only when strictly cheaper than the
// shifted pair).  costs[node] becomes the cost of the winner, like Nodes[index].Cost.
template <typename T>
__global__ __launch_bounds__(256) void k_siwt_select_level(T *__restrict__ costs, uint8_t *__restrict__ choice, int64_t NN,
                                                           int j, int d, int L, int64_t node_j, int64_t node_j1,
                                                           int64_t items)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= items) return;
    const int64_t slot0 = e >> j, idx = e & (((int64_t)1 << j) - 1);
    const int m1 = j + 1 > d ? j + 1 - d : 0, m0 = j > d ? j - d : 0;
    const int64_t shift = slot0 << m0;
    T *c = costs + (int64_t)blockIdx.y * NN;
    uint8_t *ch = choice + (int64_t)blockIdx.y * NN;
    uint8_t pick = 0;
    if (j < L && (shift & (((int64_t)1 << m1) - 1)) == 0) {            // the node was decomposed (both ways)
        const int64_t su = shift >> m1, ss = (shift + ((int64_t)1 << j)) >> m1;
        const T *c1 = c + node_j1;
        const T cu = (T)(c1[(su << (j + 1)) + 2 * idx] + c1[(su << (j + 1)) + 2 * idx + 1]);
        const T cs = (T)(c1[(ss << (j + 1)) + 2 * idx] + c1[(ss << (j + 1)) + 2 * idx + 1]);
        const T cn = c[node_j + e];
        if (!(cn < cu && cn < cs)) {
            if (cu < cs) { pick = 1; c[node_j + e] = cu; }
            else { pick = 2; c[node_j + e] = cs; }
        }
    }
    ch[node_j + e] = pick;
}

// status of the nodes of depth j1 = j + 1 from their parents' (root: in the tree)
__global__ __launch_bounds__(256) void k_siwt_mark_level(const uint8_t *__restrict__ choice, uint8_t *__restrict__ status,
                                                         int64_t NN, int j1, int d, int64_t node_j, int64_t node_j1,
                                                         int64_t items)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= items) return;
    const uint8_t *ch = choice + (int64_t)blockIdx.y * NN;
    uint8_t *stt = status + (int64_t)blockIdx.y * NN;
    if (j1 == 0) { stt[0] = 1 + ch[0]; return; }
    const int j = j1 - 1;
    const int64_t slot1 = e >> j1, idx1 = e & (((int64_t)1 << j1) - 1);
    const int m1 = j1 > d ? j1 - d : 0, m0 = j > d ? j - d : 0;
    const int64_t shift1 = slot1 << m1;
    const int s = (int)((shift1 >> j) & 1);
    const int64_t slot0 = (shift1 & (((int64_t)1 << j) - 1)) >> m0;
    const uint8_t ps = stt[node_j + (slot0 << j) + (idx1 >> 1)];
    stt[node_j1 + e] = (ps == (s ? 3 : 2)) ? (uint8_t)(1 + ch[node_j1 + e]) : (uint8_t)0;
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

int check_dims(int64_t n, int L, int d, int64_t batch, bool need_d)
{
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "siwt: bad dimensions");
    WX_REQUIRE(0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT, "@assert 0 <= L <= maxtransformlevels(x) (SIWT.jl:62)");
    if (need_d) WX_REQUIRE(1 <= d && d <= L, WX_EASSERT, "@assert 1 <= d <= L (SIWT.jl:63)");
    else WX_REQUIRE(0 <= d && d <= L, WX_EARG, "siwt: 0 <= d <= L");
    WX_REQUIRE(L <= 30 && n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "siwt: signal too long");
    return WX_OK;
}

inline unsigned blocks_for(int64_t items) { return (unsigned)((items + 255) / 256); }

template <typename T>
int api_siwpd(const T *x, T *W, T *costs, int64_t n, int L, int d, int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    if ((rc = check_dims(n, L, d, batch, true))) return rc;
    const WxSiwtGeom g = siwt_geom(L, d);
    const int64_t NS = g.coloff[L + 1], NN = g.nodeoff[L + 1];
    WX_REQUIRE((n / 2) * ((int64_t)1 << (L < d ? L : d)) / 256 < ((int64_t)1 << 31), WX_EUNSUPPORTED, "siwpd: level too large for one launch");
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dW = (T *)io.out(W, sizeof(T) * n * NS * batch);
    T *dc = costs ? (T *)io.out(costs, sizeof(T) * NN * batch) : nullptr;
    if (!dx || !dW || (costs && !dc)) return io.finish(WX_EHIP);
    // column 0 of every signal = the signal
    if (hipMemcpy2DAsync(dW, sizeof(T) * n * NS, dx, sizeof(T) * n, sizeof(T) * n, batch, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return io.finish(wx_set_error(WX_EHIP, "siwpd: copying the signals into the table failed"));
    T *dn = nullptr;
    if (dc) { dn = (T *)scr.alloc(sizeof(T) * batch); if (!dn) return io.finish(WX_EHIP); }
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {                   // gridDim.y limit
        const unsigned bc = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        T *Wb = dW + b0 * NS * n;
        if (dc) hipLaunchKernelGGL(k_siwt_norms<T>, dim3(bc), dim3(256), 0, st, (const T *)Wb, (int)n, NS * n, dn + b0);
        // costs of the depths whose nodes have at most 256 samples come out of the level that creates them
        const bool dyadic = (n & (n - 1)) == 0;
        int jc = 0;                                                 // depths 0 .. jc keep the separate cost kernel
        for (int j = 0; j < L; ++j) {
            const int64_t items = (n / 2) * (g.coloff[j + 2] - g.coloff[j + 1]);
            const bool fuse = dc && dyadic && (n >> (j + 1)) <= 256 && (n >> (j + 1)) >= 1;
            void (*kf)(T *, int, int64_t, int, int, int64_t, int64_t, int64_t, WxFilt, const T *, T *, int64_t, int64_t);
            switch (filt.F) {
#define WX_CASE(FF) case FF: kf = fuse ? k_siwt_fwd_level<T, true, FF> : k_siwt_fwd_level<T, false, FF>; break;
                WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
            default: kf = fuse ? k_siwt_fwd_level<T, true, 0> : k_siwt_fwd_level<T, false, 0>;
            }
            hipLaunchKernelGGL(kf, dim3(blocks_for(items), bc), dim3(256), 0, st, Wb, (int)n, NS, j, d, g.coloff[j],
                               g.coloff[j + 1], items, filt, fuse ? (const T *)(dn + b0) : (const T *)nullptr,
                               fuse ? dc + b0 * NN : (T *)nullptr, NN, g.nodeoff[j + 1]);
            if (!fuse) jc = j + 1;
        }
        if (dc)
            hipLaunchKernelGGL(k_siwt_costs<T>, dim3((unsigned)g.coloff[jc + 1], bc), dim3(256), 0, st, (const T *)Wb,
                               (const T *)(dn + b0), (int)n, NS, NN, g, dc + b0 * NN);
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "siwpd kernels failed to launch"));
    return io.finish(WX_OK);
}

template <typename T>
int api_siwt_bestbasis(T *costs, uint8_t *status, int L, int d, int64_t batch, void *stream)
{
    WX_REQUIRE(costs && status, WX_EARG, "NULL argument");
    WX_REQUIRE(0 <= L && L <= 30 && 0 <= d && d <= L && batch >= 0, WX_EARG, "siwt best basis: bad dimensions");
    const WxSiwtGeom g = siwt_geom(L, d);
    const int64_t NN = g.nodeoff[L + 1];
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    T *dc = (T *)io.in(costs, sizeof(T) * NN * batch);
    for (auto &it : io.items) if (it.user == costs) it.copy_out = true;      // costs are updated like Nodes[index].Cost
    uint8_t *ds = (uint8_t *)io.out(status, (size_t)NN * batch);
    if (!dc || !ds) return io.finish(WX_EHIP);
    uint8_t *dch = (uint8_t *)scr.alloc((size_t)NN * batch);
    if (!dch) return io.finish(WX_EHIP);
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned bc = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        for (int j = L; j >= 0; --j) {
            const int64_t items = g.nodeoff[j + 1] - g.nodeoff[j];
            hipLaunchKernelGGL(k_siwt_select_level<T>, dim3(blocks_for(items), bc), dim3(256), 0, st, dc + b0 * NN, dch + b0 * NN,
                               NN, j, d, L, g.nodeoff[j], g.nodeoff[j + 1], items);
        }
        for (int j1 = 0; j1 <= L; ++j1) {
            const int64_t items = g.nodeoff[j1 + 1] - g.nodeoff[j1];
            hipLaunchKernelGGL(k_siwt_mark_level, dim3(blocks_for(items), bc), dim3(256), 0, st, (const uint8_t *)(dch + b0 * NN),
                               ds + b0 * NN, NN, j1, d, j1 ? g.nodeoff[j1 - 1] : (int64_t)0, g.nodeoff[j1], items);
        }
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "siwt best-basis kernels failed to launch"));
    return io.finish(WX_OK);
}

template <typename T>
int api_isiwpd(T *W, const uint8_t *status, T *xh, int64_t n, int L, int d, int64_t batch, const double *qmf, int F,
               int literal, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(status, WX_EARG, "NULL argument");
    if ((rc = check_dims(n, L, d, batch, false))) return rc;
    const WxSiwtGeom g = siwt_geom(L, d);
    const int64_t NS = g.coloff[L + 1], NN = g.nodeoff[L + 1];
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    T *dW = (T *)io.in(W, sizeof(T) * n * NS * batch);
    for (auto &it : io.items) if (it.user == W) it.copy_out = true;          // nodes are overwritten like the reference's
    const uint8_t *ds = (const uint8_t *)io.in(status, (size_t)NN * batch);
    T *dx = (T *)io.out(xh, sizeof(T) * n * batch);
    if (!dW || !ds || !dx) return io.finish(WX_EHIP);
    WxScratch scr(st);
    const int mapn = (1 << (L + 1)) - 1;                              // one entry per (depth, IndexAtDepth)
    int *dmap = (int *)scr.alloc(sizeof(int) * (size_t)mapn * batch);
    if (!dmap) return io.finish(WX_EHIP);
    if (hipMemsetAsync(dmap, 0xFF, sizeof(int) * (size_t)mapn * batch, st) != hipSuccess)
        return io.finish(wx_set_error(WX_EHIP, "isiwpd: memset"));
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned bc = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        hipLaunchKernelGGL(k_siwt_build_map, dim3(blocks_for(NN), bc), dim3(256), 0, st, ds + b0 * NN, NN, g,
                           dmap + b0 * mapn, mapn);
        for (int j = L - 1; j >= 0; --j) {
            void (*ki)(T *, const int *, int, int, int64_t, int, int, int64_t, int64_t, WxFilt, int);
            switch (filt.F) {
#define WX_CASE(FF) case FF: ki = k_siwt_inv_level<T, FF>; break;
                WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
            default: ki = k_siwt_inv_level<T, 0>;
            }
            hipLaunchKernelGGL(ki, dim3(blocks_for(n / 2), bc), dim3(256), 0, st, dW + b0 * NS * n,
                               (const int *)(dmap + b0 * mapn), mapn, (int)n, NS, j, d, g.coloff[j], g.coloff[j + 1], filt,
                               literal ? 1 : 0);
        }
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "isiwpd kernels failed to launch"));
    if (hipMemcpy2DAsync(dx, sizeof(T) * n, dW, sizeof(T) * n * NS, sizeof(T) * n, batch, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return io.finish(wx_set_error(WX_EHIP, "isiwpd: copying the roots out failed"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {

int64_t wx_siwt_ncols(int L, int d)
{
    if (L < 0 || L > 30 || d < 0 || d > L) return -1;
    return siwt_geom(L, d).coloff[L + 1];
}
int64_t wx_siwt_nnodes(int L, int d)
{
    if (L < 0 || L > 30 || d < 0 || d > L) return -1;
    return siwt_geom(L, d).nodeoff[L + 1];
}
int wx_siwpd_f64(const double *x, double *W, double *costs, int64_t n, int L, int d, int64_t batch, const double *qmf, int F,
                 void *stream)
{ return api_siwpd<double>(x, W, costs, n, L, d, batch, qmf, F, stream); }
int wx_siwpd_f32(const float *x, float *W, float *costs, int64_t n, int L, int d, int64_t batch, const double *qmf, int F,
                 void *stream)
{ return api_siwpd<float>(x, W, costs, n, L, d, batch, qmf, F, stream); }
int wx_siwt_bestbasis_f64(double *costs, uint8_t *status, int L, int d, int64_t batch, void *stream)
{ return api_siwt_bestbasis<double>(costs, status, L, d, batch, stream); }
int wx_siwt_bestbasis_f32(float *costs, uint8_t *status, int L, int d, int64_t batch, void *stream)
{ return api_siwt_bestbasis<float>(costs, status, L, d, batch, stream); }
int wx_isiwpd_f64(double *W, const uint8_t *status, double *xh, int64_t n, int L, int d, int64_t batch, const double *qmf,
                  int F, int literal, void *stream)
{ return api_isiwpd<double>(W, status, xh, n, L, d, batch, qmf, F, literal, stream); }
int wx_isiwpd_f32(float *W, const uint8_t *status, float *xh, int64_t n, int L, int d, int64_t batch, const double *qmf, int F,
                  int literal, void *stream)
{ return api_isiwpd<float>(W, status, xh, n, L, d, batch, qmf, F, literal, stream); }

}  // extern "C"
