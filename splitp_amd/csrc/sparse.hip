// Sparse route: flattening + split score of one split entirely inside one workgroup's LDS.
//
// Replaces, like the dense route, splitp/constructions.py:31-55 + splitp/phylogenetics.py:280-300 (and is
// the device form of the reference's own sparse scorer, phylogenetics.py:303-312: top-4 singular values of
// the sparse flattening + its Frobenius norm).  A flattening of a 100 k-site alignment has ~8 k non-zeros
// in up to 10^6 cells, so instead of materialising the matrix (dense route: scatter -> Gram -> eigen, all
// through HBM) the workgroup keeps the D non-zeros as CSC + CSR lists in LDS and runs block subspace
// iteration on the implicit Gram operator:
//       W = C^T V          (sparse, CSC: one 4-lane group per column, whole waves for heavy columns)
//       Y = C W            (sparse, CSR)
//   after EACH half product the Gram matrix of the fresh block (v_mfma_f64_4x4x4) gives a Ritz sum - its trace, the
//   input block being orthonormal - and the Cholesky factor that re-orthonormalises the block (Cholesky-QR);
//   the sums converge to the sum of the 4 largest squared singular values of C by (sigma_5 / sigma_4)^2 per half product
//   score = sqrt(max(0, 1 - top4 / trace)),  trace = sum of count^2 (exact integer).
// Block width 4: on these matrices lambda_5..lambda_16 are of one magnitude, so guard vectors 5-8 buy almost
// nothing (rate lambda_5/lambda_4 ~ 2e-3 vs lambda_9/lambda_4), while a 4-wide block halves every LDS array.
// Small row sides (R <= 64, i.e. |A| <= 3 taxa) have a long column side whose W would not fit, but their
// Gram matrix does: it is accumulated exactly (u64 LDS atomics over the pairs inside every column) and the
// iteration runs on it densely.
//
// Everything is deterministic: the lists are ordered by a bitmap-rank construction (no atomic append), sums
// run in a fixed order, heavy columns are reduced by a fixed shuffle tree, the Gram accumulation is integer.
// A split whose lists / blocks do not fit the 160 KiB of LDS, or that has not converged after SPK_MAXIT
// products, is flagged (status bit 1) and re-scored by the caller on the dense route.
#include "common.h"

#define SPK_THREADS 512
#define SPK_WAVES 8
#define SPK_NB 4
#define SPK_VP 5            // row pitch of V / Y in doubles
#ifndef SPK_HEAVY
#define SPK_HEAVY 48
#endif
//        // a column / row with more entries than this is handled by a whole wave
#define SPK_MAXIT 40
#define SPK_LDS_BYTES 163840
#define SPK_SMALL_R 64

#ifdef SPK_STAMPS
__device__ long long g_spk_stamps[64];
#define SSTAMP(i)                                                                              \
    do {                                                                                       \
        __syncthreads();                                                                       \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_spk_stamps[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
extern "C" int sp_debug_spk_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spk_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : 2;
}
#else
#define SSTAMP(i)
#endif

struct SpkShared {
    double red[SPK_WAVES * 16];
    double S[16];        // X^T X of the current block (full symmetric 4 x 4, row-major)
    double top4;
    unsigned long long trace;
    int R, Kc, nheavy_c, nheavy_r, flag, pad;
    int shifts[32];
    unsigned int scan[SPK_WAVES + 1];
    unsigned int bucket[68];
};

__device__ __forceinline__ double spk_hash(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (double)x * (2.0 / 4294967296.0) - 1.0;
}

__device__ __forceinline__ double spk_rsqrt(double x) {
    double y = (double)__frsqrt_rn((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

__device__ __forceinline__ void spk_rowcol(u64 key, const int* shifts, int nr, int nc, u32& r, u32& c) {
    u32 rr = 0, cc = 0;
    for (int i = 0; i < nr; ++i) rr = (rr << 2) | (u32)((key >> shifts[i]) & 3ull);
    for (int i = 0; i < nc; ++i) cc = (cc << 2) | (u32)((key >> shifts[nr + i]) & 3ull);
    r = rr;
    c = cc;
}

// exclusive scan of one u32 per thread over the 512-thread block; returns the exclusive prefix, total in `total`
__device__ __forceinline__ u32 spk_scan(u32 v, SpkShared& sh, u32& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) sh.scan[w] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int i = 0; i < SPK_WAVES; ++i) {
        if (i < w) base += sh.scan[i];
        tot += sh.scan[i];
    }
    total = tot;
    return base + x - v;
}

// rank of key k in a presence bitmap with per-word exclusive popcount prefixes
__device__ __forceinline__ u32 bm_rank(const u64* bm, const u32* pf, u32 k) {
    return pf[k >> 6] + __popcll(bm[k >> 6] & ((1ull << (k & 63)) - 1));
}

// S = X^T X of the rows x 4 block X (row pitch `pitch` doubles) on the matrix cores: one v_mfma_f64_4x4x4 (4 blocks of
// 4 x 4 x 4) consumes 16 rows; lane l supplies X[base + l/4][l%4] as BOTH operands (A[i][k] of block b sits in lane
// i + 4b + 16k, B[k][j] in lane j + 4b + 16k - probed, tools/mfma_f64_4x4_probe.hip), D[i][j] of block b comes back in
// lane j + 4b + 16i.  Blocks are summed with two shuffles, waves through LDS in a fixed order.  Ends with a barrier.
__device__ __forceinline__ void spk_gram(const double* X, int rows, int pitch, SpkShared& sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 3, rl = lane >> 2;
    double acc = 0.0;
    for (int base = w * 16; base < rows; base += SPK_WAVES * 16) {
        const int row = base + rl;
        const double x = row < rows ? X[row * pitch + c] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, acc, 0, 0, 0);
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if ((lane & 12) == 0) sh.red[w * 16 + (lane >> 4) * 4 + (lane & 3)] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0;
        for (int i = 0; i < SPK_WAVES; ++i) t += sh.red[i * 16 + threadIdx.x];
        sh.S[threadIdx.x] = t;
    }
    __syncthreads();
}

// Cholesky-QR step: S = L L^T (sh.S, every thread redundantly in registers), X <- X L^-T by forward substitution per
// row, so that X^T X = I.  A pivot below 1e-28 of the largest diagonal marks a dead direction (rank < 4): its column
// becomes zero and stays zero.  Returns min pivot / max pivot (conditioning indicator).  Ends with a barrier.
__device__ __forceinline__ double spk_chol_apply(double* X, int rows, int pitch, const SpkShared& sh) {
    const double s00 = sh.S[0], s10 = sh.S[4], s20 = sh.S[8], s30 = sh.S[12];
    const double s11 = sh.S[5], s21 = sh.S[9], s31 = sh.S[13], s22 = sh.S[10], s32 = sh.S[14], s33 = sh.S[15];
    const double dmax = fmax(fmax(s00, s11), fmax(s22, s33));
    const double tiny = 1e-28 * dmax;
    double pmin = dmax;
    const double d0 = s00;
    const double i0 = d0 > tiny ? spk_rsqrt(d0) : 0.0;
    const double l10 = s10 * i0, l20 = s20 * i0, l30 = s30 * i0;
    const double d1 = fma(-l10, l10, s11);
    const double i1 = d1 > tiny ? spk_rsqrt(d1) : 0.0;
    const double l21 = fma(-l20, l10, s21) * i1, l31 = fma(-l30, l10, s31) * i1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, s22));
    const double i2 = d2 > tiny ? spk_rsqrt(d2) : 0.0;
    const double l32 = fma(-l31, l21, fma(-l30, l20, s32)) * i2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33)));
    const double i3 = d3 > tiny ? spk_rsqrt(d3) : 0.0;
    pmin = fmin(fmin(d0 > tiny ? d0 : dmax, d1 > tiny ? d1 : dmax), fmin(d2 > tiny ? d2 : dmax, d3 > tiny ? d3 : dmax));
    for (int row = threadIdx.x; row < rows; row += SPK_THREADS) {
        double* x = X + row * pitch;
        const double v0 = x[0] * i0;
        const double v1 = fma(-l10, v0, x[1]) * i1;
        const double v2 = fma(-l21, v1, fma(-l20, v0, x[2])) * i2;
        const double v3 = fma(-l32, v2, fma(-l31, v1, fma(-l30, v0, x[3]))) * i3;
        x[0] = v0; x[1] = v1; x[2] = v2; x[3] = v3;
    }
    __syncthreads();
    return dmax > 0 ? pmin / dmax : 1.0;
}

// Orthonormalise the block in place (its Gram matrix is already in sh.S).  One Cholesky-QR pass leaves
// |X^T X - I| ~ eps * cond(S); count flattenings have four leading singular values of one magnitude (cond < 100), so
// one pass is enough; an ill-conditioned block (pivot ratio < 0.05) gets up to two more passes (CholeskyQR2/3).
__device__ __forceinline__ void spk_orth(double* X, int rows, int pitch, SpkShared& sh) {
    double ratio = spk_chol_apply(X, rows, pitch, sh);
    for (int pass = 0; pass < 2 && ratio < 0.05; ++pass) {
        spk_gram(X, rows, pitch, sh);
        ratio = spk_chol_apply(X, rows, pitch, sh);
    }
}

// Build one list grouped by `major` (CSC: compact column, CSR: compact row) as a STABLE counting sort of the
// table order: the table is cut into SPK_WAVES contiguous chunks, wave w owns chunk w and a private row of
// per-group counters (BITS-wide fields packed into 32-bit LDS words); position of an entry = ptr[group] + (entries of
// the group in earlier chunks) + (rank among the wave's own earlier entries).  The last term is the value returned by
// the wave's own LDS atomic add: lanes of one ds_add_rtn instruction that hit the same word are resolved by the LDS in
// a fixed order, and no other wave touches that counter row, so the layout - and with it every later summation
// order - is reproducible run to run (checked by the bitwise-repeat tests).  No occupancy bitmaps, no chunking.
//   ptr[nmajor + 1] (u16), ent[D] (u32 = minor | count << 16), perm[nmajor] = groups sorted by size (descending, so
//   that the 16 four-lane groups of a wave work on groups of similar length); *nheavy = #groups > SPK_HEAVY.
template <bool MAJOR_IS_COL, int BITS>
__device__ __forceinline__ void spk_build_list(const u32* pc, const unsigned short* cnt, int D, int nmajor,
                                               unsigned short* ptr, u32* ent, unsigned short* perm, int* nheavy,
                                               u32* cw, SpkShared& sh, unsigned short* group_of = nullptr) {
    constexpr int PER = 32 / BITS;                     // counters per word
    constexpr u32 FMASK = (1u << BITS) - 1;
    const int stride = (nmajor + PER - 1) / PER;       // words per wave row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int chunk = ((D + SPK_WAVES * 64 - 1) / (SPK_WAVES * 64)) * 64;
    const int lo = min(D, w * chunk), hi = min(D, lo + chunk);
    for (int i = threadIdx.x; i < SPK_WAVES * stride; i += SPK_THREADS) cw[i] = 0;
    if (threadIdx.x < 68) sh.bucket[threadIdx.x] = 0;
    __syncthreads();
    u32* myrow = cw + w * stride;
    // lane l walks the contiguous sub-chunk [lo + l*q, lo + (l+1)*q): consecutive table entries share their leading
    // digits (hence often their row or column), so giving them to ONE lane keeps the 64 lanes of an atomic on
    // different counters (measured: 64-way same-word conflicts otherwise)
    const int q = chunk / 64;
    for (int t = 0; t < q; ++t) {                         // pass A: per-chunk group sizes
        const int i = lo + lane * q + t;
        if (i >= hi) continue;
        const u32 v = pc[i];
        const int mj = MAJOR_IS_COL ? (int)(v & 0xFFFF) : (int)(v >> 16);
        atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
    }
    __syncthreads();
    // exclusive prefix over the chunks, one thread per packed word (PER groups); totals -> ptr
    for (int q = threadIdx.x; q < stride; q += SPK_THREADS) {
        u32 run[PER];
#pragma unroll
        for (int f = 0; f < PER; ++f) run[f] = 0;
        for (int ww = 0; ww < SPK_WAVES; ++ww) {
            const u32 word = cw[ww * stride + q];
            u32 outw = 0;
#pragma unroll
            for (int f = 0; f < PER; ++f) {
                outw |= run[f] << (BITS * f);
                run[f] += (word >> (BITS * f)) & FMASK;
            }
            cw[ww * stride + q] = outw;
        }
#pragma unroll
        for (int f = 0; f < PER; ++f)
            if (q * PER + f < nmajor) ptr[q * PER + f] = (unsigned short)run[f];
    }
    __syncthreads();
    {   // exclusive scan of the group sizes -> ptr ; size buckets for the permutation
        const int per = (nmajor + SPK_THREADS - 1) / SPK_THREADS;
        const int l0 = min(nmajor, (int)threadIdx.x * per), h0 = min(nmajor, l0 + per);
        u32 sum = 0;
        for (int i = l0; i < h0; ++i) {
            const u32 c = ptr[i];
            sum += c;
            atomicAdd(&sh.bucket[c > 64 ? 64 : c], 1u);
        }
        u32 tot;
        u32 run = spk_scan(sum, sh, tot);
        for (int i = l0; i < h0; ++i) {
            const u32 c = ptr[i];
            ptr[i] = (unsigned short)run;
            run += c;
        }
        if (threadIdx.x == 0) ptr[nmajor] = (unsigned short)tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {   // descending bucket offsets; groups above SPK_HEAVY come first
        u32 run = 0, heavy = 0;
        for (int bkt = 64; bkt >= 0; --bkt) {
            const u32 c = sh.bucket[bkt];
            sh.bucket[bkt] = run;
            run += c;
            if (bkt > SPK_HEAVY) heavy = run;
        }
        *nheavy = (int)heavy;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < nmajor; m += SPK_THREADS) {
        const int c = ptr[m + 1] - ptr[m];
        perm[atomicAdd(&sh.bucket[c > 64 ? 64 : c], 1u)] = (unsigned short)m;   // order inside a bucket is irrelevant
    }
    for (int t = 0; t < q; ++t) {                         // pass B: placement (same walk as pass A)
        const int i = lo + lane * q + t;
        if (i >= hi) continue;
        const u32 v = pc[i];
        const int mj = MAJOR_IS_COL ? (int)(v & 0xFFFF) : (int)(v >> 16);
        const int mn = MAJOR_IS_COL ? (int)(v >> 16) : (int)(v & 0xFFFF);
        const u32 old = atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
        const int pos = ptr[mj] + (int)((old >> (BITS * (mj % PER))) & FMASK);
        ent[pos] = (u32)mn | ((u32)cnt[i] << 16);
        if (group_of) group_of[pos] = (unsigned short)mj;
    }
    __syncthreads();
}

// out[m][0..3] = sum over the entries e of major group m of count_e * in[minor_e][0..3].
// Light groups: one 4-lane group each; heavy groups (> SPK_HEAVY entries, listed in heavy[]): one wave each,
// 16 sub-groups striding the entries, fixed shuffle-tree reduction.  Four independent accumulation chains per
// lane keep four LDS round trips in flight (the loop is pure LDS latency otherwise); they are combined in a
// fixed order, so the result is reproducible.  Ends with a barrier.
__device__ __forceinline__ double spk_term(const u32* ent, int e, const double* in, int in_pitch, int j) {
    const u32 v = ent[e];
    return (double)(v >> 16) * in[(v & 0xFFFFu) * in_pitch + j];
}
__device__ __forceinline__ double spk_acc(double acc, const u32* ent, int e, const double* in, int in_pitch, int j) {
    const u32 v = ent[e];
    return fma((double)(v >> 16), in[(v & 0xFFFFu) * in_pitch + j], acc);   // explicit fma: the build has contraction off
}

__device__ __forceinline__ void spk_spmm(const unsigned short* ptr, const u32* ent, int nmajor,
                                         const unsigned short* perm, int nheavy, const double* in, int in_pitch,
                                         double* out, int out_pitch, int stamp_at = -1) {
    const int j = threadIdx.x & 3, g = threadIdx.x >> 2;
    for (int idx = nheavy + g; idx < nmajor; idx += SPK_THREADS / 4) {
        const int m = perm[idx];
        const int p0 = ptr[m], p1 = ptr[m + 1];
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int e = p0;
        for (; e + 3 < p1; e += 4) {
            a0 = spk_acc(a0, ent, e, in, in_pitch, j);
            a1 = spk_acc(a1, ent, e + 1, in, in_pitch, j);
            a2 = spk_acc(a2, ent, e + 2, in, in_pitch, j);
            a3 = spk_acc(a3, ent, e + 3, in, in_pitch, j);
        }
        if (e < p1) a0 = spk_acc(a0, ent, e, in, in_pitch, j);
        if (e + 1 < p1) a1 = spk_acc(a1, ent, e + 1, in, in_pitch, j);
        if (e + 2 < p1) a2 = spk_acc(a2, ent, e + 2, in, in_pitch, j);
        out[m * out_pitch + j] = (a0 + a1) + (a2 + a3);
    }
#ifdef SPK_STAMPS
    if (stamp_at >= 0) SSTAMP(stamp_at);
#endif
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sg = lane >> 2;
    for (int h = w; h < nheavy; h += SPK_WAVES) {
        const int m = perm[h];
        const int p0 = ptr[m], p1 = ptr[m + 1];
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int e = p0 + sg;
        for (; e + 48 < p1; e += 64) {
            a0 = spk_acc(a0, ent, e, in, in_pitch, j);
            a1 = spk_acc(a1, ent, e + 16, in, in_pitch, j);
            a2 = spk_acc(a2, ent, e + 32, in, in_pitch, j);
            a3 = spk_acc(a3, ent, e + 48, in, in_pitch, j);
        }
        if (e < p1) a0 = spk_acc(a0, ent, e, in, in_pitch, j);
        if (e + 16 < p1) a1 = spk_acc(a1, ent, e + 16, in, in_pitch, j);
        if (e + 32 < p1) a2 = spk_acc(a2, ent, e + 32, in, in_pitch, j);
        double acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 4) out[m * out_pitch + j] = acc;
    }
    __syncthreads();
}

__device__ __forceinline__ bool spk_converged(double s4, int it, double& prev_sum, double& prev_delta) {
    bool conv = false;
    const double delta = fabs(s4 - prev_sum);
    if (it >= 2) {
        double ratio = prev_delta > 0 ? delta / prev_delta : 0.0;
        ratio = fmin(fmax(ratio, 0.0), 0.9999);
        const double tail = delta * ratio / (1.0 - ratio);
        // (the Ritz sum itself carries ~2e-15 of rounding noise: a change below 2e-14 is at the floor)
        // tolerance: the score is sqrt(1 - s4 / trace); 1e-13 relative in s4 is < 1e-11 in any score >= 0.005
        if (it >= 3 && (delta <= 2e-14 * s4 || tail <= 1e-13 * s4)) conv = true;
    }
    prev_delta = delta;
    prev_sum = s4;
    return conv;
}

// status: bit 0 = iteration cap hit (score written but flagged), bit 1 = not handled here (re-score on the
// dense route), bits 8.. = number of operator applications.

// grid = n_al * S workgroups: block b scores split order[b / n_al] of alignment b % n_al (heaviest splits of every
// alignment first); score / status index = alignment * S + split.
__global__ __launch_bounds__(SPK_THREADS) void k_sparse_score(const AlDesc* __restrict__ als, int n_al, int n,
                                                              const SplitDev* __restrict__ splits,
                                                              const int* __restrict__ order, int S,
                                                              double* __restrict__ scores_all,
                                                              int* __restrict__ status_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SpkShared& sh = *reinterpret_cast<SpkShared*>(smem);
    const int ai = blockIdx.x % n_al;
    const int sid = order[blockIdx.x / n_al];
    const u64* __restrict__ keys = als[ai].keys;
    const u32* __restrict__ counts = als[ai].counts;
    const int64_t D = als[ai].D;
    double* __restrict__ scores = scores_all + (int64_t)ai * S;
    int* __restrict__ status = status_all + (int64_t)ai * S;
    const SplitDev& sp = splits[sid];
    const int nr = sp.nr, nc = sp.nc, rw = sp.rw, cw = sp.cw;
    if (threadIdx.x < 32) {
        const int t = threadIdx.x < nr + nc ? sp.taxa[threadIdx.x] : 0;
        sh.shifts[threadIdx.x] = 2 * (n - 1 - t);
    }
    if (threadIdx.x == 0) sh.flag = 0;
    __syncthreads();  // shifts are read by every wave below
    size_t off = (sizeof(SpkShared) + 15) & ~(size_t)15;
    auto carve = [&](size_t bytes) {
        unsigned char* p = smem + off;
        off = (off + bytes + 15) & ~(size_t)15;
        return p;
    };
    // ---- stage the table in LDS once: cell[i] = (row key << 2 nc) | col key, cnt[i] = count -----------------------
    // (every later pass reads LDS; a pass over global memory costs D / 512 serialised load latencies)
    const int W = rw + cw;
    const int Di = (int)D;
    // a-priori bounds min(4^side, D) on the compact sizes (the actual R, Kc are known only after ranking)
    const int kc_cap = (int)min((long long)D, nc >= 8 ? (long long)D : (1ll << (2 * nc)));
    const int r_cap = (int)min((long long)D, nr >= 8 ? (long long)D : (1ll << (2 * nr)));
    const bool small_sure = r_cap <= SPK_SMALL_R;   // then no CSR list is needed
    const size_t need_build = off + (size_t)D * (small_sure ? 4 : 8) + (size_t)D * 6 + (size_t)W * 12 + 256;
    if (D > 65535 || n > 16 || need_build + 2048 > SPK_LDS_BYTES) {
        if (threadIdx.x == 0) {
            scores[sid] = 0.0;
            status[sid] = 2;
        }
        return;
    }
    SSTAMP(0);
    // region A (persistent): CSC list, CSR list, pointers, heavy lists.  Region B: staging + bitmaps while
    // building, then V and W (or G).  Sizes that depend on R / Kc are carved after the ranks are known.
    u32* csc_ent = reinterpret_cast<u32*>(carve((size_t)D * 4));
    u32* csr_ent = small_sure ? nullptr : reinterpret_cast<u32*>(carve((size_t)D * 4));
    const size_t off_after_lists = off;
    u32* pc = reinterpret_cast<u32*>(carve((size_t)D * 4));
    unsigned short* cnt = reinterpret_cast<unsigned short*>(carve((size_t)D * 2));
    u64* bm = reinterpret_cast<u64*>(carve((size_t)W * 8));
    u32* pf = reinterpret_cast<u32*>(carve((size_t)W * 4));
    const int* shifts = sh.shifts;
    for (int i = threadIdx.x; i < W; i += SPK_THREADS) bm[i] = 0;
    unsigned long long tr = 0;
    // cell = sum_t digit_t(key) << dst_t: source / destination shifts are wave-uniform -> scalar registers
    int ssrc[16], sdst[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int valid = t < nr + nc;
        ssrc[t] = __builtin_amdgcn_readfirstlane(valid ? shifts[t] : 0);
        sdst[t] = __builtin_amdgcn_readfirstlane(valid ? (t < nr ? 2 * (nc + nr - 1 - t) : 2 * (nr + nc - 1 - t)) : 0);
    }
    const int ntax = nr + nc;
    for (int i = threadIdx.x; i < Di; i += SPK_THREADS) {  // no atomics in this loop: the loads pipeline
        const u64 key = keys[i];
        u32 cell = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < ntax) cell |= (u32)((key >> ssrc[t]) & 3ull) << sdst[t];
        pc[i] = cell;
        const u32 v = counts[i];
        cnt[i] = (unsigned short)v;
        tr += (unsigned long long)v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long*>(sh.red)[threadIdx.x >> 6] = tr;
    const u32 cmask = (nc >= 16) ? 0xFFFFFFFFu : ((1u << (2 * nc)) - 1);
    for (int i = threadIdx.x; i < Di; i += SPK_THREADS) {
        const u32 cell = pc[i];
        const u32 r = cell >> (2 * nc), c = cell & cmask;
        const u64 rb = 1ull << (r & 63), cb = 1ull << (c & 63);
        if (!(*(volatile u64*)(bm + (r >> 6)) & rb)) atomicOr(bm + (r >> 6), rb);
        if (!(*(volatile u64*)(bm + rw + (c >> 6)) & cb)) atomicOr(bm + rw + (c >> 6), cb);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < SPK_WAVES; ++i) t += reinterpret_cast<unsigned long long*>(sh.red)[i];
        sh.trace = t;
    }
    int dimsRC[2];
    for (int which = 0; which < 2; ++which) {
        const int base = which ? rw : 0, cntw = which ? cw : rw;
        const int per = (cntw + SPK_THREADS - 1) / SPK_THREADS;
        const int lo = min(cntw, (int)threadIdx.x * per), hi = min(cntw, lo + per);
        u32 s = 0;
        for (int i = lo; i < hi; ++i) s += __popcll(bm[base + i]);
        u32 tot;
        u32 run = spk_scan(s, sh, tot);
        for (int i = lo; i < hi; ++i) {
            pf[base + i] = run;
            run += __popcll(bm[base + i]);
        }
        dimsRC[which] = (int)tot;
    }
    __syncthreads();
    SSTAMP(1);
    const int R = dimsRC[0], Kc = dimsRC[1];
    const double trace = (double)sh.trace;
    if (R <= 4 || !(trace > 0)) {
        // min(shape) <= 4: the reference computes 1 - x/x = 0 exactly; all-zero table: 0/0 = nan
        if (threadIdx.x == 0) {
            scores[sid] = trace > 0 ? 0.0 : __builtin_nan("");
            status[sid] = 0;
        }
        return;
    }
    // compact coordinates in place: pc[i] = rr << 16 | cc
    for (int i = threadIdx.x; i < Di; i += SPK_THREADS) {
        const u32 cell = pc[i];
        const u32 r = cell >> (2 * nc), c = cell & cmask;
        pc[i] = (bm_rank(bm, pf, r) << 16) | bm_rank(bm + rw, pf + rw, c);
    }
    __syncthreads();
    const bool small = small_sure;   // (R <= 64 with a larger bound simply takes the general path)
    const int Rp = (R + 3) & ~3;
    // pointer + permutation arrays (now that R and Kc are known) are carved top-down from the end of LDS; the
    // key bitmaps are dead: the counting-sort counters start where they were
    size_t top = SPK_LDS_BYTES;
    auto carve_top = [&](size_t bytes) {
        top = (top - bytes) & ~(size_t)15;
        return smem + top;
    };
    unsigned short* csc_ptr = reinterpret_cast<unsigned short*>(carve_top((size_t)(Kc + 1) * 2));
    unsigned short* csr_ptr = reinterpret_cast<unsigned short*>(carve_top((size_t)(R + 1) * 2));
    unsigned short* perm_c = reinterpret_cast<unsigned short*>(carve_top((size_t)Kc * 2));
    unsigned short* perm_r = reinterpret_cast<unsigned short*>(carve_top((size_t)R * 2));
    const size_t build_end = reinterpret_cast<unsigned char*>(bm) - smem;
    // W row pitch: 5 doubles when LDS allows it (rows start on 32 different bank offsets instead of 8: the gathers
    // of Y = C W hit random rows), 4 otherwise
    size_t top_probe = top;
    const size_t base_iter = off_after_lists + (size_t)Rp * SPK_VP * 8 + 16;
    const int wp = (!small && base_iter + (size_t)Kc * 5 * 8 <= top_probe) ? 5 : 4;
    const size_t need_iter = base_iter + (small ? (size_t)R * R * 8 : (size_t)Kc * wp * 8);
    // counters: SPK_WAVES rows of 16-bit (8-bit when a group cannot exceed 255 entries) fields
    const bool bits8 = R <= 255 && small;
    const size_t cw_c = (size_t)SPK_WAVES * ((Kc + (bits8 ? 3 : 1)) / (bits8 ? 4 : 2)) * 4;
    const size_t cw_r = small ? 0 : (size_t)SPK_WAVES * ((R + 1) / 2) * 4;
    if (need_iter > top || build_end + (cw_c > cw_r ? cw_c : cw_r) > top || Kc > 65535) {
        if (threadIdx.x == 0) {
            scores[sid] = 0.0;
            status[sid] = 2;
        }
        return;
    }
    u32* cwbuf = reinterpret_cast<u32*>(smem + build_end);
    // small path: column of every CSC position (for the entry-parallel Gram below); it lives in the W / G area's tail
    unsigned short* colof = nullptr;
    if (small) {
        colof = reinterpret_cast<unsigned short*>(carve_top((size_t)Di * 2));
        if (need_iter > top || build_end + cw_c > top) {
            if (threadIdx.x == 0) {
                scores[sid] = 0.0;
                status[sid] = 2;
            }
            return;
        }
    }
    if (bits8)
        spk_build_list<true, 8>(pc, cnt, Di, Kc, csc_ptr, csc_ent, perm_c, &sh.nheavy_c, cwbuf, sh, colof);
    else
        spk_build_list<true, 16>(pc, cnt, Di, Kc, csc_ptr, csc_ent, perm_c, &sh.nheavy_c, cwbuf, sh);
    SSTAMP(2);
    if (!small) spk_build_list<false, 16>(pc, cnt, Di, R, csr_ptr, csr_ent, perm_r, &sh.nheavy_r, cwbuf, sh);
    SSTAMP(3);
    // ---- start block: unit vectors on the rows of the 4 largest counts (distinct rows) ----------------------------
    // (the dominant singular vectors of a count flattening sit on the few very frequent patterns); chosen by four
    // rounds of a block arg-max over (count, index), deterministic tie-break.
    int top_row[SPK_NB];
    {
        // each lane keeps the best (count, index) of its strided entries; each wave extracts its 4 best lane
        // candidates with distinct rows (shuffles only); wave 0 picks the 4 best distinct rows of the 32 candidates.
        // Candidate = (count << 32) | (0xFFFFFFFF - index): max = largest count, lowest index.  Any 4 strong distinct
        // rows make a good start block; exact ties / a lane holding two of the top rows only cost a bit of start quality.
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(sh.red);   // SPK_WAVES * 4 entries
        int* rows_out = reinterpret_cast<int*>(sh.S);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        unsigned long long mine = 0;
        for (int i = threadIdx.x; i < Di; i += SPK_THREADS) {
            const unsigned long long cand = ((unsigned long long)cnt[i] << 32) | (0xFFFFFFFFull - (unsigned)i);
            mine = cand > mine ? cand : mine;
        }
        int myrow = mine ? (int)(pc[(int)(0xFFFFFFFFull - (mine & 0xFFFFFFFFull))] >> 16) : -1;
        for (int k = 0; k < SPK_NB; ++k) {
            unsigned long long best = mine;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long o = __shfl_xor(best, d, 64);
                best = o > best ? o : best;
            }
            const int brow = best ? (int)(pc[(int)(0xFFFFFFFFull - (best & 0xFFFFFFFFull))] >> 16) : -1;
            if (lane == 0) slot[w * SPK_NB + k] = best;
            if (myrow == brow) mine = 0;   // this row is taken: drop every lane candidate on it
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            unsigned long long c = lane < SPK_WAVES * SPK_NB ? slot[lane] : 0;
            int crow = c ? (int)(pc[(int)(0xFFFFFFFFull - (c & 0xFFFFFFFFull))] >> 16) : -1;
            for (int k = 0; k < SPK_NB; ++k) {
                unsigned long long best = c;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const unsigned long long o = __shfl_xor(best, d, 64);
                    best = o > best ? o : best;
                }
                const int brow = best ? (int)(pc[(int)(0xFFFFFFFFull - (best & 0xFFFFFFFFull))] >> 16) : -1;
                if (lane == 0) rows_out[k] = brow;
                if (crow == brow) c = 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPK_NB; ++k) top_row[k] = rows_out[k];
        __syncthreads();
    }
    SSTAMP(4);
    // V and W / G are laid out over the (now dead) staging area
    off = off_after_lists;
    double* V = reinterpret_cast<double*>(carve((size_t)Rp * SPK_VP * 8));
    double* Wb = reinterpret_cast<double*>(smem + off);   // large: W (Kc x 4);  small: G (R x R)
    if (small) {
        unsigned long long* G64 = reinterpret_cast<unsigned long long*>(Wb);
        for (int i = threadIdx.x; i < R * R; i += SPK_THREADS) G64[i] = 0;
        __syncthreads();
        // exact Gram: all pairs (a <= b) inside every column.  One thread per CSC position a, walking the rest of its
        // column (<= R <= 64 steps); consecutive positions belong to consecutive threads, so a long column is spread
        // over many lanes.  u64 LDS atomics: integer, hence exact and order independent.
        for (int a = threadIdx.x; a < Di; a += SPK_THREADS) {
            const u32 va = csc_ent[a];
            const unsigned long long ca = va >> 16;
            const int ra = va & 0xFFFF;
            const int p1 = csc_ptr[colof[a] + 1];
            for (int b = a; b < p1; ++b) {
                const u32 vb = csc_ent[b];
                const int rb = vb & 0xFFFF;   // entries of a column are in table order, not row order
                atomicAdd(&G64[min(ra, rb) * R + max(ra, rb)], ca * (unsigned long long)(vb >> 16));
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < R * R; i += SPK_THREADS) {  // exact integer -> fp64, in place
            const int r = i / R, c = i % R;
            if (r > c) continue;
            G64[r * R + c] = (unsigned long long)__double_as_longlong((double)G64[r * R + c]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < R * R; i += SPK_THREADS) {
            const int r = i / R, c = i % R;
            if (r > c) Wb[r * R + c] = Wb[c * R + r];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < Rp * SPK_VP; e += SPK_THREADS) V[e] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += SPK_THREADS) {
#pragma unroll
        for (int k = 0; k < SPK_NB; ++k) V[i * SPK_VP + k] = 0.02 * spk_hash(i, k) + (top_row[k] == i ? 1.0 : 0.0);
    }
    __syncthreads();
    SSTAMP(5);
    SSTAMP(6);
    // ---- iteration: alternate half products, one Ritz sum per half product -------------------------------------
    //   h odd :  W = C^T V  (V orthonormal)  ->  trace(W^T W) = trace(V^T C C^T V) = Ritz sum of C C^T on span(V)
    //   h even:  Y = C W    (W orthonormal)  ->  trace(Y^T Y) = Ritz sum of C^T C on span(W)
    // Both are Rayleigh-Ritz sums of the same four squared singular values, each better than the last by
    // (sigma_5 / sigma_4)^2; the block is re-orthonormalised by one Cholesky-QR step on the Gram matrix that has just
    // been formed (no eigen-decomposition, no polar factor).  Small row side: dense G instead of the two sparse halves.
    double prev_sum = 0, prev_delta = 0, top4 = 0;
    int it = 0, conv = 0;
    spk_gram(V, R, SPK_VP, sh);
    spk_orth(V, R, SPK_VP, sh);
    if (small) {
        for (it = 1; it <= SPK_MAXIT; ++it) {
            // Y = G V densely (R <= 64): thread (row, j); Ritz sum = trace(V^T Y); Y staged in registers, written over V
            const int row = threadIdx.x >> 2, j = threadIdx.x & 3;
            double acc = 0, part = 0;
            if (row < R) {
                for (int k = 0; k < R; ++k) acc = fma(Wb[row * R + k], V[k * SPK_VP + j], acc);
                part = acc * V[row * SPK_VP + j];
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
            __syncthreads();
            if (row < R) V[row * SPK_VP + j] = acc;
            if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = part;
            __syncthreads();
            top4 = ((sh.red[0] + sh.red[1]) + (sh.red[2] + sh.red[3])) + ((sh.red[4] + sh.red[5]) + (sh.red[6] + sh.red[7]));
            if (spk_converged(top4, it, prev_sum, prev_delta)) {
                conv = 1;
                break;
            }
            spk_gram(V, R, SPK_VP, sh);
            spk_orth(V, R, SPK_VP, sh);
        }
    } else {
        SSTAMP(7);
        for (it = 1; it <= 2 * SPK_MAXIT; ++it) {
            double* X;
            int rows, pitch;
            if (it & 1) {
                spk_spmm(csc_ptr, csc_ent, Kc, perm_c, sh.nheavy_c, V, SPK_VP, Wb, wp, it == 1 ? 20 : -1);   // W = C^T V
                X = Wb; rows = Kc; pitch = wp;
            } else {
                spk_spmm(csr_ptr, csr_ent, R, perm_r, sh.nheavy_r, Wb, wp, V, SPK_VP, it == 2 ? 21 : -1);    // Y = C W
                X = V; rows = R; pitch = SPK_VP;
            }
            if (it <= 2) SSTAMP(7 + it);
            spk_gram(X, rows, pitch, sh);
            if (it == 2) SSTAMP(40);
            top4 = (sh.S[0] + sh.S[5]) + (sh.S[10] + sh.S[15]);
            if (spk_converged(top4, it, prev_sum, prev_delta)) {
                conv = 1;
                break;
            }
            spk_orth(X, rows, pitch, sh);
            if (it == 2) SSTAMP(41);
        }
    }
    SSTAMP(11);
    if (threadIdx.x == 0) {
        if (conv) {
            const double op = 1.0 - top4 / trace;
            scores[sid] = sqrt(op > 0 ? op : 0.0);
            status[sid] = it << 8;
        } else {
            scores[sid] = 0.0;
            status[sid] = 2 | (it << 8);  // no spectral gap behind the 4th value: let the 16-wide dense route do it
        }
    }
}

int launch_sparse_score(sp_ctx* ctx, const AlDesc* als_dev, int n_al, int n_taxa, const SplitDev* splits_dev,
                        const int* order_dev, int64_t S, double* scores, int* status) {
    if (S == 0) return SP_OK;
    if (ctx->upload_ev) SP_HIP(hipStreamWaitEvent(ctx->stream, ctx->upload_ev, 0));
    PhaseScope ps(ctx, SP_PHASE_SPARSE);
    static bool attr = false;
    if (!attr) {
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sparse_score),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_BYTES));
        attr = true;
    }
    hipLaunchKernelGGL(k_sparse_score, dim3((unsigned)(S * n_al)), dim3(SPK_THREADS), SPK_LDS_BYTES, ctx->stream, als_dev,
                       n_al, n_taxa, splits_dev, order_dev, (int)S, scores, status);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
