// Direct finisher: the four largest eigenvalues of a symmetric positive semi-definite matrix G = C C^T held in HBM, by
// Householder tridiagonalisation + Sturm-count multisection - a DIRECT method (no start vector, no stop rule, nothing
// to converge), the counterpart of LAPACK's gesdd on the reference's dense path (phylogenetics.py:280-300), which answers
// for every matrix.  The block iterations of the sparse and dense routes certify their result where the spectrum has a
// gap behind the 4th value and FLAG the split otherwise (status bit 0); flagged splits end here (api.hip: finish_flagged),
// and so do generic matrices whose smaller side is beyond the 1024 rows the dense route's eigen kernels hold in LDS.
//
// Layout: G (fp64, pitch g_pitch, full symmetric storage) is overwritten.  Per matrix a workspace of 6 vectors:
// v[2], p[2] (ping-pong per step), d, e, and a few scalars.  Step k (k = 0 .. m-3) is two launches:
//   k_fin_vec   (one workgroup per matrix)  finishes w_(k-1) = p - (tau/2)(p.v) v of the previous reflector, forms column k
//               of the matrix AS UPDATED by that reflector (the rank-2 update of step k-1 is still pending on the trailing
//               block), and from it d_k, e_k and the reflector v_k, tau_k;
//   k_fin_apply (8 rows per workgroup)      applies the PENDING update A <- A - v' w'^T - w' v'^T to its rows of the trailing
//               block while it reads them for p_k = tau_k A v_k - one read and one write of the trailing block per step.
// Traffic: 16 m^3 / 3 bytes (m = 4096: 0.37 TB), 2 m launches; all sums in fixed order (bit-reproducible).  The
// eigenvalues of the tridiagonal matrix (d, e) come from Sturm counts in the ratio form (count of negative pivots of
// T - s I, LAPACK dlaebz's recurrence with its pivot guard), 4 x 256 shifts a pass, 7 passes (257^7 > 2^53).
#include <algorithm>

#include "common.h"

#define FIN_RB 8
#define FIN_AT 256
#define FIN_VT 1024
#define FIN_CH 2048   // Sturm kernel: entries of (d, e^2) staged per LDS chunk

struct FinWs {
    double* base;
    int64_t stride;   // doubles per matrix
    int cap;          // doubles per vector
};
__device__ __forceinline__ double* fin_vec(const FinWs& ws, int sid, int which) { return ws.base + (int64_t)sid * ws.stride + (int64_t)which * ws.cap; }
// which: 0,1 = v ping-pong; 2,3 = p ping-pong; 4 = d; 5 = e; 6 = scalars (tau[2])

template <int THREADS>
__device__ __forceinline__ double fin_block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(FIN_VT) void k_fin_vec(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                    double* __restrict__ grams, FinWs ws, int k) {
    __shared__ double red[FIN_VT / 64];
    const int sid = blockIdx.x;
    const SplitDev& sp = splits[sid];
    const int m = min(dims[sid].x, sp.rcap);
    if (m <= 0 || k > max(m - 2, 0)) return;
    const double* __restrict__ A = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int cur = k & 1, prv = cur ^ 1;
    double* v = fin_vec(ws, sid, cur);
    const double* vp = fin_vec(ws, sid, prv);
    double* pp = fin_vec(ws, sid, 2 + prv);
    double* d = fin_vec(ws, sid, 4);
    double* e = fin_vec(ws, sid, 5);
    double* sc = fin_vec(ws, sid, 6);
    const double taup = k > 0 ? sc[prv] : 0.0;
    const bool pend = taup != 0.0;
    if (pend) {   // w' = p' - (tau'/2)(p'.v') v' over the trailing block of step k-1 (rows k .. m-1), in place of p'
        double part = 0;
        for (int i = k + threadIdx.x; i < m; i += FIN_VT) part += pp[i] * vp[i];
        const double kk = 0.5 * taup * fin_block_sum<FIN_VT>(part, red);
        for (int i = k + threadIdx.x; i < m; i += FIN_VT) pp[i] -= kk * vp[i];
        __syncthreads();
    }
    const double vpk = pend ? vp[k] : 0.0, wpk = pend ? pp[k] : 0.0;
    const double* __restrict__ Ak = A + (int64_t)k * gp;   // row k == column k
    double tail = 0;
    for (int i = k + 1 + threadIdx.x; i < m; i += FIN_VT) {
        double c = Ak[i];
        if (pend) c -= vp[i] * wpk + pp[i] * vpk;
        v[i] = c;
        if (i > k + 1) tail += c * c;
    }
    if (threadIdx.x == 0) d[k] = pend ? Ak[k] - 2.0 * (vpk * wpk) : Ak[k];
    if (m == 1) return;
    tail = fin_block_sum<FIN_VT>(tail, red);   // (its barriers publish v[k + 1])
    const double x0 = v[k + 1];
    if (k == m - 2) {
        if (threadIdx.x == 0) {
            e[k] = x0;
            const double akk = A[(int64_t)(k + 1) * gp + k + 1];
            d[k + 1] = pend ? akk - 2.0 * (vp[k + 1] * pp[k + 1]) : akk;
            sc[cur] = 0.0;
        }
        return;
    }
    double tau = 0.0, ek = x0, v0 = x0;
    if (tail > 0) {
        const double nrm = sqrt(x0 * x0 + tail);
        const double alpha = x0 > 0 ? -nrm : nrm;
        v0 = x0 - alpha;
        tau = 2.0 / (tail + v0 * v0);
        ek = alpha;
    }
    __syncthreads();   // every thread has read v[k + 1]
    if (threadIdx.x == 0) {
        v[k + 1] = v0;
        e[k] = ek;
        sc[cur] = tau;
    }
}

__global__ __launch_bounds__(FIN_AT) void k_fin_apply(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                      double* __restrict__ grams, FinWs ws, int k) {
    __shared__ double red[FIN_RB][FIN_AT / 64];
    __shared__ double vpi[FIN_RB], wpi[FIN_RB];
    const int sid = blockIdx.y;
    const SplitDev& sp = splits[sid];
    const int m = min(dims[sid].x, sp.rcap);
    if (k > m - 3) return;
    const int row0 = k + 1 + (int)blockIdx.x * FIN_RB;
    if (row0 >= m) return;
    double* __restrict__ A = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int cur = k & 1, prv = cur ^ 1;
    const double* v = fin_vec(ws, sid, cur);
    const double* vp = fin_vec(ws, sid, prv);
    const double* wp = fin_vec(ws, sid, 2 + prv);
    double* p = fin_vec(ws, sid, 2 + cur);
    const double* sc = fin_vec(ws, sid, 6);
    const double tau = sc[cur];
    const bool pend = k > 0 && sc[prv] != 0.0;
    if (!pend && tau == 0.0) {
        if (threadIdx.x < FIN_RB && row0 + (int)threadIdx.x < m) p[row0 + threadIdx.x] = 0.0;
        return;
    }
    if (threadIdx.x < FIN_RB) {
        const int i = row0 + threadIdx.x;
        vpi[threadIdx.x] = pend && i < m ? vp[i] : 0.0;
        wpi[threadIdx.x] = pend && i < m ? wp[i] : 0.0;
    }
    __syncthreads();
    double acc[FIN_RB];
#pragma unroll
    for (int r = 0; r < FIN_RB; ++r) acc[r] = 0.0;
    const int nrow = min(FIN_RB, m - row0);
    for (int j = k + 1 + threadIdx.x; j < m; j += FIN_AT) {
        const double vj = v[j];
        const double vpj = pend ? vp[j] : 0.0, wpj = pend ? wp[j] : 0.0;
#pragma unroll
        for (int r = 0; r < FIN_RB; ++r) {
            if (r < nrow) {
                double* cell = A + (int64_t)(row0 + r) * gp + j;
                double a = *cell;
                if (pend) {
                    a -= vpi[r] * wpj + wpi[r] * vpj;
                    *cell = a;
                }
                acc[r] += a * vj;
            }
        }
    }
    const int lane = threadIdx.x & 63, w = sp_wave_id();
#pragma unroll
    for (int r = 0; r < FIN_RB; ++r) {
        double x = acc[r];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) x += __shfl_xor(x, dd, 64);
        if (lane == 0) red[r][w] = x;
    }
    __syncthreads();
    if (threadIdx.x < nrow) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < FIN_AT / 64; ++i) t += red[threadIdx.x][i];
        p[row0 + threadIdx.x] = tau * t;
    }
}

// Top-4 eigenvalues of the tridiagonal (d, e) of every matrix -> score and status.  1024 threads = 4 groups of 256
// shifts; group g brackets the (g+1)-th largest eigenvalue.  N(s) = number of eigenvalues below s = negative pivots.
__global__ __launch_bounds__(FIN_VT) void k_fin_sturm(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                      FinWs ws, const int* __restrict__ out_idx,
                                                      double* __restrict__ scores, int* __restrict__ status) {
    __shared__ double red[FIN_VT / 64];
    __shared__ double ds[FIN_CH], es[FIN_CH];
    __shared__ double lo[4], hi[4];
    __shared__ int nflag[4];
    const int sid = blockIdx.x;
    const SplitDev& sp = splits[sid];
    const int m = min(dims[sid].x, sp.rcap);
    const int oi = out_idx ? out_idx[sid] : sid;
    const double* d = fin_vec(ws, sid, 4);
    const double* e = fin_vec(ws, sid, 5);
    double tr = 0, bound = 0;
    for (int i = threadIdx.x; i < m; i += FIN_VT) {
        tr += d[i];
        const double b = fabs(d[i]) + (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < m ? fabs(e[i]) : 0.0);
        bound = fmax(bound, b);
    }
    tr = fin_block_sum<FIN_VT>(tr, red);
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) bound = fmax(bound, __shfl_xor(bound, dd, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[sp_wave_id()] = bound;
    __syncthreads();
    bound = 0;
    for (int i = 0; i < FIN_VT / 64; ++i) bound = fmax(bound, red[i]);
    const int code = 4 | (min(m, 0x7FFFFF) << 8);
    if (m <= 4 || !(tr > 0) || !(bound > 0)) {
        // min(shape) <= 4: the reference computes 1 - x/x = 0 exactly; all-zero matrix: 0/0 = nan (eigen.hip: k_eig_init)
        if (threadIdx.x == 0) {
            scores[oi] = tr > 0 ? 0.0 : __builtin_nan("");
            status[oi] = code;
        }
        return;
    }
    const double inv = 1.0 / bound;
    const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int below_max = m - (g + 1);   // N(s) <= below_max  <=>  s <= lambda_(g+1)
    if (threadIdx.x < 4) {
        lo[threadIdx.x] = -1.0;
        hi[threadIdx.x] = 1.0;
    }
    __syncthreads();
    const double pivmin = 1e-290;
    for (int pass = 0; pass < 7; ++pass) {
        const double l = lo[g], h = hi[g];
        const double s = l + (h - l) * ((double)(t + 1) * (1.0 / 257.0));
        int cnt = 0;
        double q = 1.0;
        for (int c0 = 0; c0 < m; c0 += FIN_CH) {
            __syncthreads();
            for (int i = threadIdx.x; i < FIN_CH && c0 + i < m; i += FIN_VT) {
                ds[i] = d[c0 + i] * inv;
                const double ee = c0 + i > 0 ? e[c0 + i - 1] * inv : 0.0;
                es[i] = ee * ee;
            }
            __syncthreads();
            const int n = min(FIN_CH, m - c0);
            for (int i = 0; i < n; ++i) {
                q = ds[i] - s - es[i] / q;   // (i == 0 of the first chunk: es = 0)
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0 ? 1 : 0;
            }
        }
        if (threadIdx.x < 4) nflag[threadIdx.x] = 0;
        __syncthreads();
        if (cnt <= below_max) atomicAdd(&nflag[g], 1);   // (N is monotone in s: the flagged shifts are a prefix)
        __syncthreads();
        const int nf = nflag[g];
        const double nl = nf > 0 ? l + (h - l) * ((double)nf * (1.0 / 257.0)) : l;
        const double nh = nf < 256 ? l + (h - l) * ((double)(nf + 1) * (1.0 / 257.0)) : h;
        __syncthreads();
        if (t == 0) {
            lo[g] = nl;
            hi[g] = nh;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double top4 = 0;
        for (int j = 3; j >= 0; --j) top4 += 0.5 * (lo[j] + hi[j]);
        top4 *= bound;
        const double op = 1.0 - top4 / tr;
        scores[oi] = sqrt(op > 0 ? op : 0.0);
        status[oi] = code;
    }
}

// How many doubles of workspace `launch_direct_top4` needs for n_mats matrices of at most max_rows rows.
size_t direct_ws_doubles(int64_t n_mats, int64_t max_rows) {
    const int64_t cap = round_up(std::max<int64_t>(max_rows, 1), 64);
    return (size_t)n_mats * (size_t)(6 * cap + 64);
}

// Scores of n_mats Gram matrices (grams + splits[s].g_off, pitch g_pitch, rows dims[s].x, fp64, DESTROYED) by the direct
// method; max_rows = largest dims[s].x (host knowledge: the caller has fetched the dims).  Results go to
// scores[out_idx[s]] / status[out_idx[s]] (out_idx == nullptr: s).  ws: direct_ws_doubles(n_mats, max_rows) doubles.
int launch_direct_top4(sp_ctx* ctx, const SplitDev* splits_dev, const int2* dims_dev, int64_t n_mats, int max_rows,
                       double* grams, double* ws_dev, const int* out_idx_dev, double* scores, int* status) {
    if (n_mats == 0) return SP_OK;
    SP_REQUIRE(n_mats < 65536, SP_ELIMIT, "direct solver: %lld matrices in one batch", (long long)n_mats);
    FinWs ws;
    ws.base = ws_dev;
    ws.cap = (int)round_up(std::max(max_rows, 1), 64);
    ws.stride = 6 * (int64_t)ws.cap + 64;
    PhaseScope ps(ctx, SP_PHASE_EIGEN);
    for (int k = 0; k <= std::max(max_rows - 2, 0); ++k) {
        hipLaunchKernelGGL(k_fin_vec, dim3((unsigned)n_mats), dim3(FIN_VT), 0, ctx->stream, splits_dev, dims_dev, grams, ws, k);
        if (k <= max_rows - 3) {
            const int len = max_rows - k - 1;
            hipLaunchKernelGGL(k_fin_apply, dim3((unsigned)((len + FIN_RB - 1) / FIN_RB), (unsigned)n_mats), dim3(FIN_AT), 0,
                               ctx->stream, splits_dev, dims_dev, grams, ws, k);
        }
    }
    hipLaunchKernelGGL(k_fin_sturm, dim3((unsigned)n_mats), dim3(FIN_VT), 0, ctx->stream, splits_dev, dims_dev, ws, out_idx_dev,
                       scores, status);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
