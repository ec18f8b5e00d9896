// Top-4 eigenvalues of the Gram matrix + the split score.
//
// Second half of the replacement for splitp/phylogenetics.py:280-300 (__dense_split_score):
//     score = (1 - sum(sigma[:4]^2) / sum(sigma^2)) ** 0.5
// With G = C C^T:  sum(sigma^2) = trace(G) (exact for integer counts) and sigma[:4]^2 are the four
// largest eigenvalues of G.  They are found by block orthogonal (subspace) iteration, block width 16
// (the N of v_mfma_f64_16x16x4_f64), in the G^2 form (eig_small.h):
//     Y = G V            fp64 MFMA, G streamed from L2/HBM (symmetric: read as columns, 128-B segments)
//     S = Y^T Y          = V^T G^2 V, 16 x 16 Gram by MFMA
//     S = P D P^T        one-wave 2 x 2-block Jacobi;  Ritz values of G = sqrt(D)
//     V = Y P D^-1/2     next orthonormal block (+ one Newton-Schulz polish step)
// until the sum of the four largest Ritz values stops moving (geometric-tail estimate below
// 1e-14 relative).  Phylogenetic flattenings have lambda_17 / lambda_4 < 1e-3, so this takes 3-4
// products.  The score itself is then taken from the FIRST-power projection G1 = V^T Y of the same
// subspace (one more 16 x 16 Jacobi, once per split): G^2 values lose eigenvalues below ~1e-7 lambda_1.
//
// Launch structure: the product Y = G V is the only heavy step and G (up to 1024^2 doubles per
// split) has to stream from HBM once per product, so it runs as its own kernel over ALL splits
// with one workgroup per 64 rows (k_eig_gv: the whole chip pulls on HBM), while the small per-split
// algebra (Gram of Y, Jacobi, block update on an R x 16 block held in LDS) runs one workgroup per split
// (k_eig_init, k_eig_rr).  EIG_NFAST product rounds are enqueued without host synchronisation;
// splits that have converged turn their workgroups into no-ops.  k_eig_finish then takes any
// split that is still not converged (arbitrary matrices without a spectral gap) to convergence
// inside one workgroup, capped at EIG_MAXIT (status bit 0).
#include "common.h"

#ifdef EIG_STAMPS
__device__ long long g_eig_stamps[64];
__device__ int g_eig_stamp_on;   // set by the kernel under the stamps
__device__ double g_eig_dump[256];   // projected matrix of workgroup 0 in the stamped round
#define STAMP(i)                                                                         \
    do {                                                                                 \
        __syncthreads();                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0 && g_eig_stamp_on) g_eig_stamps[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define EIG_STAMP(i) STAMP(i)
#else
#define STAMP(i)
#endif

#include "eig_small.h"

#define gram16 gram_nb<16, EIG_VP>
#define rowmul16 rowmul_nb<16, EIG_VP>
// LDS head of the per-split kernels: EigShared, then EigSpec (k_eig_rr's second Jacobi problem), then the R x 16 block
#define EIG_HEAD_BYTES ((((sizeof(EigShared) + 15) & ~(size_t)15)) + ((sizeof(EigSpec) + 15) & ~(size_t)15))
#define ritz_orth16 ritz_orth_nb<16, EIG_VP>
#define orth_near16 orth_near_nb<16, EIG_VP>

// ---- per split: trace, degenerate cases, start block -------------------------------------------------
template <typename GT>
__global__ __launch_bounds__(EIG_THREADS) void k_eig_init(const SplitDev* __restrict__ splits,
                                                          const int2* __restrict__ dims,
                                                          const GT* __restrict__ grams,
                                                          EigState* __restrict__ states, double* __restrict__ vt_pool,
                                                          double* __restrict__ scores, int* __restrict__ status,
                                                          const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + EIG_HEAD_BYTES);
    const int sid = order[blockIdx.x];  // heaviest splits first
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const int Rp = (R + 15) & ~15;
    const GT* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;

    // trace(G) = sum of all squared singular values (exact for integer counts)
    double tr = 0;
    for (int i = threadIdx.x; i < R; i += EIG_THREADS) tr += (double)G[(int64_t)i * gp + i];
    tr = block_sum(tr, sh);
    EigState st;
    st.trace = tr; st.prev_sum = 0; st.prev_delta = 0; st.prev_ratio = 1.0; st.top4 = 0; st.it = 0; st.done = 0; st.R = R; st.fp_it = 0;
    st.f_prev_sum = 0; st.f_prev_delta = 0; st.f_prev_ratio = 1.0;
    if (R <= 4 || !(tr > 0)) {
        // min(shape) <= 4: the reference computes 1 - x/x = 0 exactly; all-zero matrix: 0/0 = nan
        if (threadIdx.x == 0) {
            scores[sid] = (tr > 0) ? 0.0 : __builtin_nan("");
            status[sid] = 0;
            st.done = 1;
            states[sid] = st;
        }
        return;
    }
    if (R <= EIG_B) {
        // The whole Gram matrix fits the 16 x 16 Jacobi: eigenvalues directly, to full relative accuracy (an iteration
        // on G^2 drops eigenvalues below ~1e-7 lambda_1, which on a nearly rank-4 flattening IS the score)
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            sh.H[i * EIG_VP + j] = (i < R && j < R) ? (double)G[(int64_t)i * gp + j] : 0.0;
        }
        __syncthreads();
        top4_of_H(sh);
        if (threadIdx.x == 0) {
            st.done = 1;
            st.top4 = sh.top4;
            states[sid] = st;
            write_score(st.top4, st.trace, 1, true, scores, status, sid);
        }
        return;
    }
    // Start block: fixed pseudo-random entries (robust for any matrix) plus a strong component on the
    // rows with the largest diagonal of G - for count matrices the dominant singular vectors sit on
    // the rows of the few very frequent patterns - then Cholesky-QR x2.
    double* dg = sh.part;  // R <= 1024 doubles fit in the 8 KiB reduction buffer
    for (int i = threadIdx.x; i < Rp; i += EIG_THREADS) dg[i] = i < R ? (double)G[(int64_t)i * gp + i] : -1.0;
    __syncthreads();
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) {
        const int row = e >> 4, col = e & 15;
        V[row * EIG_VP + col] = row < R ? 0.02 * hash_unit(row, col) : 0.0;
    }
    __syncthreads();
    // (no early exit on rank >= EIG_B: with the data-dependent exit every iteration waited for its own LDS read, 2048
    // dependent reads per thread on a 1024-row side = 85 us of k_eig_init; as a plain counted loop the reads pipeline)
    for (int i = threadIdx.x; i < R; i += EIG_THREADS) {
        const double di = dg[i];
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < Rp; ++j) {      // (padding entries are -1: never ahead of a diagonal entry of a PSD matrix)
            const double dj = dg[j];
            rank += (dj > di || (dj == di && j < i)) ? 1 : 0;
        }
        if (rank < EIG_B) V[i * EIG_VP + rank] += 1.0;
    }
    __syncthreads();
    orth_near16(V, Rp, sh);  // unit columns + Newton-Schulz polish (the start block is nearly orthogonal as it stands)
    store_vt(V, Rp, vt_pool + sp.ev_off, sp.rcap);
    if (threadIdx.x == 0) states[sid] = st;
}

// ---- Y = G V for every split: one workgroup per (split, 64-row block) ---------------------------------
// Per wave one 16-row tile; per 16-k block a lane loads 4 consecutive doubles of its G row (32 B: a wave
// reads 16 rows x 128 B) and 4 consecutive doubles of its V column, and issues 4 MFMAs; MFMA j
// sums k in {k0 + 4g + j : g = 0..3}, so the 16 k's are covered exactly once.
template <typename GT>
struct G4 {};
template <>
struct G4<double> {
    static __device__ __forceinline__ void load(const double* p, double* a) {
        const double2 x = *reinterpret_cast<const double2*>(p), y = *reinterpret_cast<const double2*>(p + 2);
        a[0] = x.x; a[1] = x.y; a[2] = y.x; a[3] = y.y;
    }
};
template <>
struct G4<int> {
    static __device__ __forceinline__ void load(const int* p, double* a) {
        const int4 x = *reinterpret_cast<const int4*>(p);
        a[0] = (double)x.x; a[1] = (double)x.y; a[2] = (double)x.z; a[3] = (double)x.w;
    }
};

template <typename GT>
__global__ __launch_bounds__(256) void k_eig_gv(const GramItem* __restrict__ items,
                                                const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                const GT* __restrict__ grams,
                                                const EigState* __restrict__ states,
                                                const double* __restrict__ vt_pool, double* __restrict__ y_pool) {
    const GramItem it = items[blockIdx.x];
    const int sid = it.sid;
    if (sid < 0 || states[sid].done) return;
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const int Rp = (R + 15) & ~15;
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    const int row0 = it.ti * 64 + w * 16;
    if (row0 >= Rp) return;
    const GT* __restrict__ g = grams + sp.g_off + (int64_t)(row0 + fr) * sp.g_pitch + 4 * fk;
    const double* __restrict__ v = vt_pool + sp.ev_off + (int64_t)fr * sp.rcap + 4 * fk;
    double4_t acc = {0, 0, 0, 0};
    // 32 k's a step: a lane loads 8 consecutive entries of its G row (the 4 lanes of a row read one full 128-byte line of
    // an int32 G, 256 bytes of an fp64 G) and 8 consecutive entries of its V column; MFMA j sums k in {k0 + 8 g + j}.
    // (16 k's a step - 64-byte half lines of the int32 G - streamed at 3.3 TB/s.)  A 16-wide tail follows when needed.
    int k0 = 0;
#pragma unroll 4
    for (; k0 + 32 <= Rp; k0 += 32) {
        double a[8];
        G4<GT>::load(g + k0 + 4 * fk, a);            // g already carries 4 fk: 8 fk in all
        G4<GT>::load(g + k0 + 4 * fk + 4, a + 4);
        const double* __restrict__ vb = v + k0 + 4 * fk;
        const double2 b01 = *reinterpret_cast<const double2*>(vb), b23 = *reinterpret_cast<const double2*>(vb + 2);
        const double2 b45 = *reinterpret_cast<const double2*>(vb + 4), b67 = *reinterpret_cast<const double2*>(vb + 6);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b01.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b01.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b23.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b23.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4], b45.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[5], b45.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[6], b67.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[7], b67.y, acc, 0, 0, 0);
    }
    for (; k0 < Rp; k0 += 16) {
        double a[4];
        G4<GT>::load(g + k0, a);
        const double2 b01 = *reinterpret_cast<const double2*>(v + k0);
        const double2 b23 = *reinterpret_cast<const double2*>(v + k0 + 2);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b01.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b01.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b23.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b23.y, acc, 0, 0, 0);
    }
    double* __restrict__ y = y_pool + sp.ev_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[(int64_t)(row0 + fk + 4 * r) * EIG_B + fr] = acc[r];
}


// ---- per split: Rayleigh-Ritz on (V, Y = G V), convergence test, next orthonormal block ----------------
__global__ __launch_bounds__(EIG_THREADS) void k_eig_rr(const SplitDev* __restrict__ splits,
                                                        const int2* __restrict__ dims, EigState* __restrict__ states,
                                                        double* __restrict__ vt_pool,
                                                        const double* __restrict__ y_pool,
                                                        double* __restrict__ scores, int* __restrict__ status,
                                                        const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + EIG_HEAD_BYTES);
    const int sid = order[blockIdx.x];
    EigState st = states[sid];
    if (st.done) return;
    const SplitDev& sp = splits[sid];
    const int R = st.R;
    const int Rp = (R + 15) & ~15;
    double* __restrict__ Vt = vt_pool + sp.ev_off;
    const double* __restrict__ Y = y_pool + sp.ev_off;
    const int vp = sp.rcap;
#ifdef EIG_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        g_eig_stamp_on = (st.it == EIG_STAMP_ROUND);
        if (g_eig_stamp_on) g_eig_stamps[39] = 0;
    }
    __syncthreads();
#endif
    STAMP(0);
    // Y -> LDS working block
#pragma unroll 8
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) V[(e >> 4) * EIG_VP + (e & 15)] = Y[e];
    __syncthreads();
    STAMP(1);
    proj_first_power<EIG_VP>(V, Rp, Vt, vp, sh);  // G1 = V^T G V on the block that produced Y (kept for the final score)
    STAMP(4);
    // the (symmetrised) first-power projection goes to wave 1's Jacobi, which runs next to the G^2 Jacobi of wave 0
    EigSpec& spec = *reinterpret_cast<EigSpec*>(smem_raw + ((sizeof(EigShared) + 15) & ~(size_t)15));
    if (threadIdx.x < 256) {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        spec.H[i * EIG_VP + j] = 0.5 * (sh.G1[i * EIG_VP + j] + sh.G1[j * EIG_VP + i]);
    }
    __syncthreads();
    ritz_orth_nb<16, EIG_VP>(V, Rp, sh, &spec);  // Ritz values (sh.top4) + next orthonormal block in one pass
    STAMP(2);
    st.it += 1;
    st.top4 = sh.top4;
    const double g2_sum = sh.top4;
    bool g2_conv = update_convergence(g2_sum, st.it, st.prev_sum, st.prev_delta, st.prev_ratio);
    g2_conv = g2_conv || certified_stop(g2_sum, st.prev_delta, st.it, st.trace, sh);
#ifdef EIG_STAMPS
    if (st.it == 3 && threadIdx.x == 0 && blockIdx.x < 128) {   // what the certified stop saw (diagnostic build)
        g_eig_dump[blockIdx.x] = st.prev_delta / g2_sum;
        g_eig_dump[128 + blockIdx.x] = (st.trace - sh.sum_all) / sh.theta4;
    }
#endif
    // (the score comes from the first-power Ritz values of the same subspace; sets st.top4)
    if (accept_first_power(g2_conv, g2_sum, st, sh, &spec)) {
        STAMP(5);
        if (threadIdx.x == 0) {
            st.done = 1;
            states[sid] = st;
            write_score(st.top4, st.trace, st.it, true, scores, status, sid);
        }
        return;
    }
    store_vt(V, Rp, Vt, vp);
    STAMP(3);
    if (threadIdx.x == 0) states[sid] = st;
}

// ---- finisher: splits still not converged after the fast rounds iterate inside one workgroup ----------
template <typename GT>
__global__ __launch_bounds__(EIG_THREADS) void k_eig_finish(const SplitDev* __restrict__ splits,
                                                            const GT* __restrict__ grams,
                                                            EigState* __restrict__ states,
                                                            const double* __restrict__ vt_pool,
                                                            double* __restrict__ scores, int* __restrict__ status,
                                                            const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + EIG_HEAD_BYTES);
    const int sid = order[blockIdx.x];
    EigState st = states[sid];
    if (st.done) return;
    const SplitDev& sp = splits[sid];
    const int R = st.R;
    const int Rp = (R + 15) & ~15;
    const GT* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    load_vt(V, Rp, vt_pool + sp.ev_off, sp.rcap);
    __syncthreads();
    const int ntile = Rp >> 4;
    int converged = 0, ran = 0;
    while (st.it < EIG_MAXIT) {
        // ---- Y = G V : wave w owns row tiles w, w+8, ... ; k outer so one V fragment feeds all tiles
        double4_t acc[EIG_MAXT];
#pragma unroll
        for (int t = 0; t < EIG_MAXT; ++t) acc[t] = (double4_t){0, 0, 0, 0};
        for (int k0 = 0; k0 < Rp; k0 += 4) {
            const double b = V[(k0 + fk) * EIG_VP + fr];
            const GT* __restrict__ gk = G + (int64_t)(k0 + fk) * gp + fr;  // G[k][row] == G[row][k]
#pragma unroll
            for (int t = 0; t < EIG_MAXT; ++t) {
                const int tile = w + t * EIG_WAVES;
                if (tile < ntile) {
                    const double a = (double)gk[tile * 16];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        // G1 = V^T Y while V is still in LDS and Y in the accumulators: accumulator r of a tile holds rows 4r + fk of
        // Y - exactly the B operand of an MFMA whose 4 k's are those rows
        double4_t h1 = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < EIG_MAXT; ++t) {
            const int tile = w + t * EIG_WAVES;
            if (tile < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h1 = __builtin_amdgcn_mfma_f64_16x16x4f64(V[(tile * 16 + 4 * r + fk) * EIG_VP + fr], acc[t][r], h1,
                                                              0, 0, 0);
            }
        }
        __syncthreads();  // every wave is done reading V
#pragma unroll
        for (int t = 0; t < EIG_MAXT; ++t) {
            const int tile = w + t * EIG_WAVES;
            if (tile < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) V[(tile * 16 + fk + 4 * r) * EIG_VP + fr] = acc[t][r];
            }
        }
        reduce16(h1, sh, sh.G1);  // (starts with the barrier that publishes V)
        ritz_orth16(V, Rp, sh);
        st.it += 1;
        st.top4 = sh.top4;
        ran = 1;
        const double g2_sum = sh.top4;
        bool g2_conv = update_convergence(g2_sum, st.it, st.prev_sum, st.prev_delta, st.prev_ratio);
        g2_conv = g2_conv || certified_stop(g2_sum, st.prev_delta, st.it, st.trace, sh);
        if (accept_first_power(g2_conv, g2_sum, st, sh)) {
            converged = 1;
            break;
        }
    }
    if (ran && !converged) {   // cap hit: the flagged estimate comes from the first power as well
        first_power_top4(sh);
        st.top4 = sh.top4;
    }
    if (threadIdx.x == 0) {
        st.done = 1;
        states[sid] = st;
        write_score(st.top4, st.trace, st.it, converged != 0, scores, status, sid);
    }
}

template <typename GT>
static int launch_eigen_t(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits,
                          const int2* dims, const GT* grams, const GramItem* rowblocks_dev, int64_t n_rowblocks,
                          const int* order_dev, double* scores, int* status, int64_t n_rowblocks_a, int64_t n_splits_a) {
    int maxr = 0;
    size_t ev_elems = 0;
    for (const auto& s : splits) {
        maxr = std::max(maxr, (int)s.rcap);
        ev_elems = std::max(ev_elems, (size_t)s.ev_off + (size_t)s.rcap * EIG_B);
    }
    SP_REQUIRE(maxr <= EIG_MAXR, SP_ELIMIT,
               "eigen kernels: the smaller side of a flattening has %d (padded) rows; this build keeps the iteration "
               "block in LDS and supports at most %d (n_taxa <= 11 on the dense route)", maxr, EIG_MAXR);
    const size_t S = splits.size();
    const size_t st_bytes = (S * sizeof(EigState) + 255) & ~(size_t)255;
    SP_CHECK(ctx->eigws.ensure(st_bytes + 2 * ev_elems * sizeof(double)));
    EigState* states = ctx->eigws.as<EigState>();
    double* vt = reinterpret_cast<double*>(ctx->eigws.as<unsigned char>() + st_bytes);
    double* yp = vt + ev_elems;
    PhaseScope ps(ctx, SP_PHASE_EIGEN);
    const size_t lds_head = EIG_HEAD_BYTES;
    const size_t lds = lds_head + (size_t)maxr * EIG_VP * sizeof(double);
    static PerDeviceOnce attr;   // (per kernel type GT; the largest block: EIG_MAXR rows)
    if (attr.need(ctx->device)) {
        const size_t lds_max = lds_head + (size_t)EIG_MAXR * EIG_VP * sizeof(double);
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_init<GT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_rr), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_max));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_finish<GT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr.done(ctx->device);
    }
    // One pipeline = init, EIG_NFAST x (G V product over all row blocks, per-split Rayleigh-Ritz), finisher.  Every kernel
    // touches only the states / blocks / scores of its own splits, so two disjoint sets of splits are two independent
    // pipelines.
    auto pipeline = [&](hipStream_t stream, const int* order, size_t n_s, const GramItem* rowblocks, int64_t n_rb,
                        size_t lds_bytes) {
        if (n_s == 0) return;
        hipLaunchKernelGGL(k_eig_init<GT>, dim3((unsigned)n_s), dim3(EIG_THREADS), lds_bytes, stream, splits_dev, dims,
                           grams, states, vt, scores, status, order);
        for (int round = 0; round < EIG_NFAST; ++round) {
            hipLaunchKernelGGL(k_eig_gv<GT>, dim3((unsigned)n_rb), dim3(256), 0, stream, rowblocks, splits_dev, dims, grams,
                               states, vt, yp);
            hipLaunchKernelGGL(k_eig_rr, dim3((unsigned)n_s), dim3(EIG_THREADS), lds_bytes, stream, splits_dev, dims, states,
                               vt, yp, scores, status, order);
        }
        hipLaunchKernelGGL(k_eig_finish<GT>, dim3((unsigned)n_s), dim3(EIG_THREADS), lds_bytes, stream, splits_dev, grams,
                           states, vt, scores, status, order);
    };
    // The per-split work is one wave's 16 x 16 Jacobi whatever the size of the side, and a workgroup sized for a
    // 1024-row side (139 KB of LDS) owns its CU: 501 splits were two rounds of workgroups per kernel.  The short sides
    // (<= EIG_SMALL_ROWS rows: 48 KB, three workgroups per CU) therefore run as a pipeline of their own on the context's
    // side stream, concurrently with the long sides: one fork after the Gram kernel, one join before the scores are read.
    const bool two = n_splits_a > 0 && (size_t)n_splits_a < S && n_rowblocks_a >= 0 && ctx->opt.eigen_one_stream == 0;
    if (two && !ctx->side) {
        SP_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        SP_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        SP_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    if (two) {
        const size_t lds_small = lds_head + (size_t)EIG_SMALL_ROWS * EIG_VP * sizeof(double);
        SP_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        SP_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        pipeline(ctx->stream, order_dev, (size_t)n_splits_a, rowblocks_dev, n_rowblocks_a, lds);
        pipeline(ctx->side, order_dev + n_splits_a, S - (size_t)n_splits_a, rowblocks_dev + n_rowblocks_a,
                 n_rowblocks - n_rowblocks_a, lds_small);
        SP_HIP(hipEventRecord(ctx->ev_join, ctx->side));
        SP_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    } else {
        pipeline(ctx->stream, order_dev, S, rowblocks_dev, n_rowblocks, lds);
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}

int launch_eigen(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                 const void* grams, bool g_i32, const GramItem* rowblocks_dev, int64_t n_rowblocks,
                 const int* order_dev, double* scores, int* status, int64_t n_rowblocks_a, int64_t n_splits_a) {
    if (splits.empty()) return SP_OK;
    if (g_i32)
        return launch_eigen_t<int>(ctx, splits_dev, splits, dims, (const int*)grams, rowblocks_dev, n_rowblocks,
                                   order_dev, scores, status, n_rowblocks_a, n_splits_a);
    return launch_eigen_t<double>(ctx, splits_dev, splits, dims, (const double*)grams, rowblocks_dev, n_rowblocks,
                                  order_dev, scores, status, n_rowblocks_a, n_splits_a);
}

#ifdef EIG_STAMPS
extern "C" int sp_debug_eig_dump(double* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eig_dump), sizeof(double) * 256) == hipSuccess ? 0 : 2;
}
extern "C" int sp_debug_eig_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eig_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : 2;
}
#endif
