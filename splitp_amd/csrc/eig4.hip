// Dense route, eigen phase (round 4): top-4 eigenvalue sum of every split's Gram matrix G = C C^T (int32 or fp64, in HBM)
// by the SAME certified 4-wide block iteration the sparse route runs on its exact small-side Gram matrices
// (sparse.hip "Gram path", sparse_common.h: spk_gram / Cholesky-QR / spk_converged<dense_g>), one 1024-thread workgroup per
// split, the R x 4 block in LDS.  Second half of the replacement for LAPACK gesdd at splitp/phylogenetics.py:281-285.
//
// Why not the 16-wide block of eigen.hip (rounds 1 - 3): its per-split work is one wave's 16 x 16 Jacobi per product
// (60 - 90 us whatever the side length) behind a separate G V kernel - init 105 + 3 x (165 + 190) us at BASELINE config 2,
// 60 % of the north-star pipeline - and its stop rule is an estimate (two measured ratios).  Here a product is a stream of
// G through one workgroup (thread i owns row i and reads column i of the symmetric G, i.e. row j coalesced over the lanes;
// V[j][0..3] is an LDS broadcast), the 4 x 4 algebra is one MFMA Gram + a Cholesky factor, the start block is four columns
// of G itself (G e_r: the "free first product" of the sparse route), and the stop is CERTIFIED: rest = trace - s >= lambda_5,
// the block's smallest Ritz value <= lambda_4, so rest <= 0.6 theta_4 proves a gap and bounds the error left.  A matrix
// without such a gap is not iterated to death: it leaves flagged (status bit 1) and the host hands it to the direct solver
// (finish.hip).  Real alignments: every split of config 2 certifies at its 3rd sum.
#define SPK_THREADS 1024
#include "sparse_common.h"

#define E4_MAXIT 40
#ifndef E4_BATCH
#define E4_BATCH 16
#endif

template <typename GT>
__device__ __forceinline__ GT e4_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ int e4_load<int>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0);
}
template <>
__device__ __forceinline__ double e4_load<double>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    typedef unsigned e4_u2 __attribute__((ext_vector_type(2)));
    const e4_u2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0);
    return __hiloint2double((int)w.y, (int)w.x);
}

// Gram matrices of at most 16 rows: the one-wave 16 x 16 Jacobi (eig_small.h) gives every eigenvalue to its own relative
// accuracy, and the score is taken from the sum of the eigenvalues BEHIND the fourth - no 1 - top4 / trace cancellation.  (The
// iteration's sum is good to ~4e-15 of the trace; on a 6 x 5 table of numerical rank 4 that is 3e-14 in score^2 - scores of
// 8e-6 came back 2e-9 off in the first soak of round 4.)  A kernel of its own, one wave per split, launched BEHIND k_eig4 and
// only for the splits that need it - a score below 1e-3 (where the floor of the sum shows) or no certificate: inlined into
// k_eig4 the Jacobi cost that kernel 28 bytes of scratch per lane, and run for every short side ahead of it (one wave's
// Jacobi is 60 us of latency) it cost the north-star pipeline 0.06 ms.
template <typename GT>
__global__ __launch_bounds__(64) void k_eig4_small(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                   const GT* __restrict__ grams, double* __restrict__ scores,
                                                   int* __restrict__ status) {
    __shared__ EigShared esh;
    const int sid = blockIdx.x;
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    if (R <= 4 || R > EIG_B) return;   // (R <= 4 and the all-zero matrix: k_eig4)
    if (!(status[sid] & 3) && !(scores[sid] <= 1e-3)) return;   // certified and far above the floor of the iteration's sum
    const GT* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int lane = threadIdx.x;
    double tr = 0;
    for (int e = lane; e < EIG_B * EIG_B; e += 64) {
        const int r = e >> 4, c = e & 15;
        const double g = (r < R && c < R) ? (double)G[(int64_t)r * gp + c] : 0.0;
        esh.H[r * EIG_VP + c] = g;
        if (r == c) tr += g;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    if (!(tr > 0)) return;             // all-zero: k_eig4 writes the nan
    __syncthreads();
    jacobi_nb<EIG_B>(esh);
    const double th = lane < EIG_B ? fmax(esh.theta[lane], 0.0) : -1.0;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < EIG_B; ++j) {
        const double o = __shfl(th, j, 64);
        rank += (o > th || (o == th && j < lane)) ? 1 : 0;
    }
    double top = (lane < EIG_B && rank < 4) ? th : 0.0, rest = (lane < EIG_B && rank >= 4) ? th : 0.0;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        top += __shfl_xor(top, d, 64);
        rest += __shfl_xor(rest, d, 64);
    }
    if (lane == 0) {
        scores[sid] = sqrt(rest / (top + rest));
        status[sid] = 1 << 8;
    }
}

template <typename GT>
__global__ __launch_bounds__(SPK_THREADS) void k_eig4(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                      const GT* __restrict__ grams, double* __restrict__ scores,
                                                      int* __restrict__ status, const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SpkShared& sh = *reinterpret_cast<SpkShared*>(smem);
    double* V = reinterpret_cast<double*>(smem + ((sizeof(SpkShared) + 15) & ~(size_t)15));   // [R][4], row-major
    __shared__ unsigned long long cand_red[SPK_WAVES];
    __shared__ int top_row[4];
    const int sid = order ? order[blockIdx.x] : (int)blockIdx.x;
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const GT* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int i = threadIdx.x, lane = threadIdx.x & 63, w = spk_wave_id();
    // (descriptor from wave-uniform values only: the pointer's halves through readfirstlane, cdna_hip_programming.md T8)
    const unsigned long long gaddr = (unsigned long long)G;
    const unsigned long long gaddr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gaddr >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(gaddr & 0xFFFFFFFFull));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(gaddr_u), 0, (int)__builtin_amdgcn_readfirstlane((int)((unsigned)gp * (unsigned)sp.rcap * (unsigned)sizeof(GT))),
        0x00020000);
    const double dg = i < R ? (double)G[(int64_t)i * gp + i] : 0.0;
    double tr = dg;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    if (lane == 0) sh.red[w] = tr;
    __syncthreads();
    double trace = 0;
#pragma unroll
    for (int ww = 0; ww < SPK_WAVES; ++ww) trace += sh.red[ww];
    __syncthreads();
    if (R <= 4 || !(trace > 0)) {
        // min(shape) <= 4: the reference computes 1 - x/x = 0 exactly; all-zero matrix: 0/0 = nan (eigen.hip: k_eig_init)
        if (threadIdx.x == 0) {
            scores[sid] = trace > 0 ? 0.0 : __builtin_nan("");
            status[sid] = 0;
        }
        return;
    }
    // the 4 rows with the largest diagonal (for count matrices the dominant singular vectors sit on the rows of the few very
    // frequent patterns): candidates = (value bits, low 10 bits replaced by 1023 - row), extracted in descending order
    unsigned long long bound = ~0ull;
    for (int k = 0; k < 4; ++k) {
        unsigned long long c = 0;
        if (i < R) c = ((unsigned long long)__double_as_longlong(dg > 0 ? dg : 0.0) & ~0x3FFull) | (unsigned long long)(1023 - i);
        unsigned long long best = c < bound ? c : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) cand_red[w] = best;
        __syncthreads();
        unsigned long long b = 0;
#pragma unroll
        for (int ww = 0; ww < SPK_WAVES; ++ww) b = cand_red[ww] > b ? cand_red[ww] : b;
        if (threadIdx.x == 0) top_row[k] = 1023 - (int)(b & 0x3FFull);
        bound = b;
        __syncthreads();
    }
    // start block: columns top_row[k] of G (= G e_r, one product for free) plus 1 % noise for the directions they miss
    // (noise of 1 % of a column's NORM: the column G e_r / G[r][r] has norm ~1, R hash values of mean square 1/3)
    const double nscale = 0.01 * spk_rsqrt((double)R * (1.0 / 3.0));
    if (i < R) {
        double nrm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double gd = (double)G[(int64_t)top_row[k] * gp + top_row[k]];
            nrm[k] = gd > 0 ? 1.0 / gd : 1.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool dup = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) dup = dup || (q < k && top_row[q] == top_row[k]);   // (R >= 5: cannot happen; belt and braces)
            const double col = dup ? 0.0 : (double)G[(int64_t)top_row[k] * gp + i] * nrm[k];
            V[i * 4 + k] = col + nscale * spk_hash((unsigned)i, (unsigned)k);
        }
    }
    __syncthreads();
    spk_gram(V, R, 4, 1, sh);
    spk_chol_factor(sh);
    spk_orth(V, R, 4, 1, sh);
    double prev_sum = 0, prev_delta = 0, prev_ratio = 1.0, top4 = 0;
    int it = 0, conv = 0;
    // Sum number 1 of the sequence is free as well: the block of unit vectors e_r that the start block is the image of has
    // the Ritz sum G[r][r] summed over the four rows.  With it the rule sees its third sum - two differences, one measured
    // ratio - after TWO streamed products instead of three (the certificate itself - gap and error bound - only uses the
    // last difference, between two blocks that are one product apart).
    {
        double s0 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s0 += (double)G[(int64_t)top_row[k] * gp + top_row[k]];
        (void)spk_converged<true>(s0, 0.0, trace, 1, prev_sum, prev_delta, prev_ratio);
    }
    for (it = 1; it <= E4_MAXIT; ++it) {
        // Y = G V: thread i owns row i; G[j][i] == G[i][j], so the 64 lanes of a wave read 64 consecutive cells of row j and
        // V[j][0..3] is one broadcast read of LDS.  (Measured and dropped, round 4: four rows per thread - one V[j] read
        // feeding 16 FMAs - with the idle waves taking slices of the j range: the same 0.28 ms for the 126 5|5 splits of
        // config 2, because the phase is bound by the stream of G itself - 126 x 2.2 MB x 3 products in 0.28 ms = 3 TB/s
        // with half of the CUs pulling - not by LDS or issue.)
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (i < ((R + 63) & ~63)) {   // (whole waves; the idle lanes of the last one re-read column R - 1)
            // Row j of G through a buffer descriptor: the lane's column offset is ONE 32-bit register for every load, the row
            // offset a scalar - E4_BATCH loads in flight cost E4_BATCH data registers and no 64-bit address arithmetic (as
            // flat loads the compiler spent two vector registers and a v_lshl_add_u64 per load and waited for the first loads
            // before it issued the last).
            const unsigned voff = (unsigned)min(i, R - 1) * (unsigned)sizeof(GT);
            const unsigned rowbytes = (unsigned)gp * (unsigned)sizeof(GT);   // (R <= 1024 rows of <= 8 KB)
            int j = 0;
            for (; j + E4_BATCH <= R; j += E4_BATCH) {
                GT g[E4_BATCH];
#pragma unroll
                for (int u = 0; u < E4_BATCH; ++u) g[u] = e4_load<GT>(rsrc, voff, (unsigned)(j + u) * rowbytes);
                __builtin_amdgcn_sched_barrier(0);   // (all of the batch's requests leave before the first product)
#pragma unroll
                for (int u = 0; u < E4_BATCH; ++u) {
                    const double gd = (double)g[u];
                    const double4 v = *reinterpret_cast<const double4*>(V + (j + u) * 4);
                    a0 = fma(gd, v.x, a0);
                    a1 = fma(gd, v.y, a1);
                    a2 = fma(gd, v.z, a2);
                    a3 = fma(gd, v.w, a3);
                }
            }
            for (; j < R; ++j) {
                const double gd = (double)e4_load<GT>(rsrc, voff, (unsigned)j * rowbytes);
                const double4 v = *reinterpret_cast<const double4*>(V + j * 4);
                a0 = fma(gd, v.x, a0);
                a1 = fma(gd, v.y, a1);
                a2 = fma(gd, v.z, a2);
                a3 = fma(gd, v.w, a3);
            }
        }
        double part = 0;
        if (i < R) part = (a0 * V[i * 4] + a1 * V[i * 4 + 1]) + (a2 * V[i * 4 + 2] + a3 * V[i * 4 + 3]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        __syncthreads();   // every thread has read the whole of V
        if (i < R) {
            V[i * 4] = a0; V[i * 4 + 1] = a1; V[i * 4 + 2] = a2; V[i * 4 + 3] = a3;
        }
        if (lane == 0) sh.red[w] = part;
        __syncthreads();
        top4 = 0;
#pragma unroll
        for (int ww = 0; ww < SPK_WAVES; ++ww) top4 += sh.red[ww];   // Ritz sum = trace(V^T G V), V orthonormal
        __syncthreads();
        spk_gram(V, R, 4, 1, sh);                                    // Y^T Y = V^T G^2 V: squared Ritz values
        const double rest_s = trace - top4;
        spk_chol_factor(sh, it >= 2, rest_s > 0 ? rest_s * rest_s * (1.0 / 0.3) : 0.0);
        if (spk_converged<true>(top4, sh.L[11] > 0 ? sp_fsqrt(sh.L[11]) : 0.0, trace, it + 1, prev_sum, prev_delta, prev_ratio)) {
            conv = 1;
            break;
        }
        spk_orth(V, R, 4, 1, sh);
    }
    if (threadIdx.x == 0) {
        const double op = 1.0 - top4 / trace;
        scores[sid] = sqrt(op > 0 ? op : 0.0);
        status[sid] = conv ? (it << 8) : (2 | (it << 8));   // bit 1: no certificate - the direct solver's (finish.hip)
    }
}

int launch_eigen4(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                  const void* grams, bool g_i32, const int* order_dev, double* scores, int* status) {
    if (splits.empty()) return SP_OK;
    int maxr = 0;
    for (const auto& s : splits) maxr = std::max(maxr, (int)s.rcap);
    SP_REQUIRE(maxr <= EIG_MAXR, SP_ELIMIT,
               "eigen kernel: the smaller side of a flattening has %d (padded) rows; one workgroup owns a split (a row per "
               "thread) and takes at most %d (n_taxa <= 11 on the dense route)", maxr, EIG_MAXR);
    const size_t lds = ((sizeof(SpkShared) + 15) & ~(size_t)15) + (size_t)maxr * 4 * sizeof(double);
    static PerDeviceOnce attr;   // (the largest block the kernel can be asked for: EIG_MAXR rows)
    if (attr.need(ctx->device)) {
        const int lds_max = (int)(((sizeof(SpkShared) + 15) & ~(size_t)15) + (size_t)EIG_MAXR * 4 * sizeof(double));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig4<int>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig4<double>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        attr.done(ctx->device);
    }
    PhaseScope ps(ctx, SP_PHASE_EIGEN);
    if (g_i32)
        hipLaunchKernelGGL(k_eig4<int>, dim3((unsigned)splits.size()), dim3(SPK_THREADS), lds, ctx->stream, splits_dev, dims,
                           (const int*)grams, scores, status, order_dev);
    else
        hipLaunchKernelGGL(k_eig4<double>, dim3((unsigned)splits.size()), dim3(SPK_THREADS), lds, ctx->stream, splits_dev, dims,
                           (const double*)grams, scores, status, order_dev);
    int minr = EIG_MAXR;
    for (const auto& sp : splits) minr = std::min(minr, (int)sp.rcap);
    if (minr <= 64) {   // (a side of <= 16 compact rows has rcap = 64: only then can the small kernel have anything to do)
        if (g_i32)
            hipLaunchKernelGGL(k_eig4_small<int>, dim3((unsigned)splits.size()), dim3(64), 0, ctx->stream, splits_dev, dims,
                               (const int*)grams, scores, status);
        else
            hipLaunchKernelGGL(k_eig4_small<double>, dim3((unsigned)splits.size()), dim3(64), 0, ctx->stream, splits_dev, dims,
                               (const double*)grams, scores, status);
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}
