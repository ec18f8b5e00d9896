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
//       S = Y^T Y = V^T G^2 V,  S = P D P^T (4 x 4 Jacobi),  V <- Y P D^-1/2,  Newton-Schulz polish
//       sum_i sqrt(D_i) -> sum of the 4 largest singular values^2 of C (converged when it stops moving)
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
    double red[SPK_WAVES * 12];
    double S[12];        // 10 unique entries of the symmetric 4 x 4 (row-major upper: 00 01 02 03 11 12 13 22 23 33)
    double T[16];        // 4 x 4 transform applied to the block
    double Zprev[16];    // S0^-1/2 of the previous product (warm start of the inverse-square-root iteration)
    int have_z, padz;
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

// Sum the 10 unique entries of X^T X (X: R x 4 block, pitch SPK_VP) into sh.S.  Fixed reduction tree.
__device__ __forceinline__ void spk_gram(const double* X, int R, SpkShared& sh) {
    double s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = 0;
    for (int row = threadIdx.x; row < R; row += SPK_THREADS) {
        const double a = X[row * SPK_VP], b = X[row * SPK_VP + 1], c = X[row * SPK_VP + 2], d = X[row * SPK_VP + 3];
        s[0] = fma(a, a, s[0]); s[1] = fma(a, b, s[1]); s[2] = fma(a, c, s[2]); s[3] = fma(a, d, s[3]);
        s[4] = fma(b, b, s[4]); s[5] = fma(b, c, s[5]); s[6] = fma(b, d, s[6]);
        s[7] = fma(c, c, s[7]); s[8] = fma(c, d, s[8]); s[9] = fma(d, d, s[9]);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[i] += __shfl_xor(s[i], d, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) sh.red[w * 12 + i] = s[i];
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double t = 0;
        for (int i = 0; i < SPK_WAVES; ++i) t += sh.red[i * 12 + threadIdx.x];
        sh.S[threadIdx.x] = t;
    }
    __syncthreads();
}

// X <- X * T (T 4 x 4 row-major in sh.T), row-wise.  Ends with a barrier.
__device__ __forceinline__ void spk_apply(double* X, int R, const SpkShared& sh) {
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = sh.T[i];
    for (int row = threadIdx.x; row < R; row += SPK_THREADS) {
        const double a = X[row * SPK_VP], b = X[row * SPK_VP + 1], c = X[row * SPK_VP + 2], d = X[row * SPK_VP + 3];
        X[row * SPK_VP + 0] = a * t[0] + b * t[4] + c * t[8] + d * t[12];
        X[row * SPK_VP + 1] = a * t[1] + b * t[5] + c * t[9] + d * t[13];
        X[row * SPK_VP + 2] = a * t[2] + b * t[6] + c * t[10] + d * t[14];
        X[row * SPK_VP + 3] = a * t[3] + b * t[7] + c * t[11] + d * t[15];
    }
    __syncthreads();
}

// 4 x 4 symmetric Jacobi in registers (cyclic order), executed by wave 0 (all lanes redundantly):
// S (sh.S) = P D P^T.  Writes T = P D^-1/2 (dead directions zeroed) and sh.top4 = sum sqrt(D_i).
__device__ __forceinline__ void spk_jacobi4(SpkShared& sh) {
    if (threadIdx.x < 64) {
        double a[4][4], p[4][4];
        a[0][0] = sh.S[0]; a[0][1] = sh.S[1]; a[0][2] = sh.S[2]; a[0][3] = sh.S[3];
        a[1][1] = sh.S[4]; a[1][2] = sh.S[5]; a[1][3] = sh.S[6];
        a[2][2] = sh.S[7]; a[2][3] = sh.S[8]; a[3][3] = sh.S[9];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < i) a[i][j] = a[j][i];
                p[i][j] = i == j ? 1.0 : 0.0;
            }
        for (int sweep = 0; sweep < 12; ++sweep) {
            double rel = 0;
            const double dmx = fmax(fmax(fabs(a[0][0]), fabs(a[1][1])), fmax(fabs(a[2][2]), fabs(a[3][3])));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i + 1; j < 4; ++j) {
                    const double v = a[i][j], dd = fabs(a[i][i] * a[j][j]);
                    if (v * v > 1e-40 * dmx * dmx) rel = fmax(rel, dd > 0 ? v * v / dd : 1.0);
                }
            if (!(rel > 1e-22)) break;
#pragma unroll
            for (int ip = 0; ip < 3; ++ip)
#pragma unroll
                for (int iq = ip + 1; iq < 4; ++iq) {
                    const double app = a[ip][ip], aqq = a[iq][iq], apq = a[ip][iq];
                    double c = 1.0, s = 0.0;
                    if (apq != 0.0 && apq * apq > 1e-40 * fabs(app * aqq)) {
                        // angle in f32 (cheap); c = rsqrt(1 + t^2), s = t c in fp64: orthogonal to fp64 accuracy for
                        // any t, an inexact angle only leaves a ~1e-7 |apq| residue for the next sweep
                        const float num = (float)(aqq - app), den = 2.0f * (float)apq;
                        float tf;
                        if (fabsf(num) > 1e18f * fabsf(den)) {
                            tf = den / (2.0f * num);
                        } else {
                            const float tau = num / den;
                            tf = (tau >= 0.f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
                        }
                        const double t = (double)tf;
                        c = spk_rsqrt(1.0 + t * t);
                        s = t * c;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // columns p, q of A and P
                        const double akp = a[k][ip], akq = a[k][iq];
                        a[k][ip] = c * akp - s * akq;
                        a[k][iq] = s * akp + c * akq;
                        const double pkp = p[k][ip], pkq = p[k][iq];
                        p[k][ip] = c * pkp - s * pkq;
                        p[k][iq] = s * pkp + c * pkq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // rows p, q of A
                        const double apk = a[ip][k], aqk = a[iq][k];
                        a[ip][k] = c * apk - s * aqk;
                        a[iq][k] = s * apk + c * aqk;
                    }
                }
        }
        const double dmax = fmax(fmax(a[0][0], a[1][1]), fmax(a[2][2], a[3][3]));
        double top = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dj = a[j][j];
            const bool alive = dj > 1e-28 * dmax && dj > 0;
            const double rj = alive ? spk_rsqrt(dj) : 0.0;
            top += alive ? sqrt(dj) : 0.0;
            if (threadIdx.x == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sh.T[i * 4 + j] = p[i][j] * rj;
            }
        }
        if (threadIdx.x == 0) sh.top4 = top;
    }
    __syncthreads();
}

// 4 x 4 helpers on row-major arrays (every thread computes them redundantly from LDS broadcasts: no serial
// single-wave section, no extra barriers)
__device__ __forceinline__ void spk_sym_from(const double* S10, double (&s)[16]) {
    s[0] = S10[0]; s[1] = S10[1]; s[2] = S10[2]; s[3] = S10[3];
    s[4] = S10[1]; s[5] = S10[4]; s[6] = S10[5]; s[7] = S10[6];
    s[8] = S10[2]; s[9] = S10[5]; s[10] = S10[7]; s[11] = S10[8];
    s[12] = S10[3]; s[13] = S10[6]; s[14] = S10[8]; s[15] = S10[9];
}
__device__ __forceinline__ void spk_apply_reg(double* X, int R, const double (&t)[16]) {
    for (int row = threadIdx.x; row < R; row += SPK_THREADS) {
        const double a = X[row * SPK_VP], b = X[row * SPK_VP + 1], c = X[row * SPK_VP + 2], d = X[row * SPK_VP + 3];
        X[row * SPK_VP + 0] = a * t[0] + b * t[4] + c * t[8] + d * t[12];
        X[row * SPK_VP + 1] = a * t[1] + b * t[5] + c * t[9] + d * t[13];
        X[row * SPK_VP + 2] = a * t[2] + b * t[6] + c * t[10] + d * t[14];
        X[row * SPK_VP + 3] = a * t[3] + b * t[7] + c * t[11] + d * t[15];
    }
    __syncthreads();
}

// 4 x 4 matrix product c = a b (row-major), registers
__device__ __forceinline__ void mm4(const double (&a)[16], const double (&b)[16], double (&c)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c[4 * i + j] = fma(a[4 * i], b[j], fma(a[4 * i + 1], b[4 + j], fma(a[4 * i + 2], b[8 + j], a[4 * i + 3] * b[12 + j])));
}

// Ritz values + next orthonormal block from Y (held in X), in place.
//   Polar decomposition Y = U H:  U = Y (Y^T Y)^-1/2 is orthonormal and H = (Y^T Y)^1/2, so trace(H) =
//   sum_i sqrt(eig_i(Y^T Y)) is exactly the sum of the four Ritz values and U is the next block.
//   S0 = Y^T Y (one block reduction);  Z = S0^-1/2 by the coupled Newton-Schulz iteration on the 4 x 4 matrix itself
//   (registers of wave 0: A <- A (3I - Z A)/2, Z <- (3I - Z A) Z / 2 from A = c S0, Z = I with c = 1 / Gershgorin
//   bound, quadratic);  X <- X Z sqrt(c);  one Newton-Schulz polish step on the block removes the rounding of the 4 x 4
//   solve;  trace(H) = trace(T^T S0) with T the total transform.
//   Taken when Y is well conditioned (column norms within a factor 7, scaled off-diagonals <= 0.25) - true from the
//   first product on for count flattenings, whose four leading singular values are of one magnitude.
//   Otherwise (arbitrary blocks / matrices): Jacobi eigen-decomposition of S0, X <- X P D^-1/2, then the same polish.
__device__ __forceinline__ void spk_ritz_orth(double* X, int R, SpkShared& sh, int st0 = -1) {
    if (st0 >= 0) SSTAMP(st0);
    spk_gram(X, R, sh);
    if (st0 >= 0) SSTAMP(st0 + 1);
    double s0[16], t[16];
    spk_sym_from(sh.S, s0);
    const double dmax = fmax(fmax(s0[0], s0[5]), fmax(s0[10], s0[15]));
    double d[4];
    bool alive[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        alive[i] = s0[5 * i] > 1e-28 * dmax && s0[5 * i] > 0;
        d[i] = alive[i] ? spk_rsqrt(s0[5 * i]) : 0.0;
    }
    double offmax = 0, dmin = dmax;
    bool all_alive = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        all_alive = all_alive && alive[i];
        dmin = fmin(dmin, s0[5 * i]);
#pragma unroll
        for (int j = i + 1; j < 4; ++j) offmax = fmax(offmax, fabs(s0[4 * i + j]) * d[i] * d[j]);
    }
    double top_jacobi = -1.0;
    __syncthreads();
    if (all_alive && offmax <= 0.25 && dmin >= 0.02 * dmax) {
        if (threadIdx.x < 64) {   // wave 0: Z = S0^-1/2, all lanes redundantly in registers
            double z[16], m[16], tmp[16];
            // Newton iteration Z <- Z (3I - Z S0 Z) / 2 (locally quadratic).  Warm start: once the block has settled in
            // the invariant subspace, S0 = V^T G^2 V barely changes from one product to the next, so the previous
            // inverse square root is already accurate to ~1e-5 and two steps finish it.  Cold start: c I with
            // c = 1 / sqrt(Gershgorin bound) (every eigenvalue of c^2 S0 in (0, 1]).
            bool warm = sh.have_z != 0;
            if (warm) {
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = sh.Zprev[i];
                mm4(z, s0, tmp);
                mm4(tmp, z, m);
                double err = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) err = fmax(err, fabs(m[i] - ((i % 5 == 0) ? 1.0 : 0.0)));
                warm = err < 0.5;
            }
            if (!warm) {
                double gb = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    gb = fmax(gb, fabs(s0[4 * i]) + fabs(s0[4 * i + 1]) + fabs(s0[4 * i + 2]) + fabs(s0[4 * i + 3]));
                const double c = spk_rsqrt(gb);
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = (i % 5 == 0) ? c : 0.0;
            }
            // every pass measures m = Z S0 Z and applies the Newton factor (1.5 I - 0.5 m); the pass whose MEASURED
            // defect is already <= 1e-8 leaves a defect of ~1e-16 behind (the step squares it), so it is the last one
            // and doubles as the polish: no second block reduction is needed on this path.
            double prev_err = 1e300;
            for (int iter = 0; iter < 12; ++iter) {
                mm4(z, s0, tmp);
                mm4(tmp, z, m);                    // m = Z S0 Z -> I
                double err = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    err = fmax(err, fabs(m[i] - ((i % 5 == 0) ? 1.0 : 0.0)));
                    m[i] = ((i % 5 == 0) ? 1.5 : 0.0) - 0.5 * m[i];
                }
                mm4(z, m, tmp);
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = tmp[i];
                if (err <= 1e-8 || err >= prev_err) break;
                prev_err = err;
            }
            if (threadIdx.x == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sh.T[i] = z[i];
                    sh.Zprev[i] = z[i];
                }
                sh.have_z = 1;
            }
        }
        __syncthreads();
    } else {
        spk_jacobi4(sh);   // writes sh.T = P D^-1/2 and sh.top4
        top_jacobi = sh.top4;
    }
    if (st0 >= 0) SSTAMP(st0 + 2);
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = sh.T[i];
    spk_apply_reg(X, R, t);
    if (st0 >= 0) SSTAMP(st0 + 3);
    for (int iter = 0; iter < 12 && top_jacobi >= 0; ++iter) {   // block polish: robust (Jacobi) path only
        spk_gram(X, R, sh);
        double sk[16];
        spk_sym_from(sh.S, sk);
        double err = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double target = (i == j && alive[i] && sk[5 * i] > 0.25) ? 1.0 : 0.0;
                err = fmax(err, fabs(sk[4 * i + j] - target));
            }
        if (err <= 2e-15) break;
        double m[16], tn[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[4 * i + j] = (i == j ? 1.5 : 0.0) - 0.5 * sk[4 * i + j];
        mm4(t, m, tn);
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = tn[i];
        spk_apply_reg(X, R, m);
        if (err * err <= 1e-17) break;   // the step just applied squares the defect: below fp64 resolution
    }
    if (st0 >= 0) SSTAMP(st0 + 4);
    // sum of the Ritz values = trace(T^T S0)   (Jacobi path: S0's eigenvalues were computed directly)
    double tr = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tr += t[i] * s0[i];
    __syncthreads();
    if (threadIdx.x == 0) sh.top4 = top_jacobi >= 0 ? top_jacobi : tr;
    __syncthreads();
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
        if (it >= 3 && (delta <= 2e-14 * s4 || tail <= 1e-14 * s4)) conv = true;
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
    if (threadIdx.x == 0) {
        sh.flag = 0;
        sh.have_z = 0;
    }
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
    // ---- iteration ------------------------------------------------------------------------------------------
    double prev_sum = 0, prev_delta = 0, top4 = 0;
    int it = 0, conv = 0;
    for (it = 1; it <= SPK_MAXIT; ++it) {
        if (small) {
            // Y = G V densely (R <= 64): thread (row, j); Y staged in registers, then written over V
            const int row = threadIdx.x >> 2, j = threadIdx.x & 3;
            double acc = 0;
            if (row < R)
                for (int k = 0; k < R; ++k) acc += Wb[row * R + k] * V[k * SPK_VP + j];
            __syncthreads();
            if (row < R) V[row * SPK_VP + j] = acc;
            __syncthreads();
        } else {
            if (it == 1) SSTAMP(7);
            spk_spmm(csc_ptr, csc_ent, Kc, perm_c, sh.nheavy_c, V, SPK_VP, Wb, wp, it == 1 ? 20 : -1);   // W = C^T V
            if (it == 1) SSTAMP(8);
            spk_spmm(csr_ptr, csr_ent, R, perm_r, sh.nheavy_r, Wb, wp, V, SPK_VP, it == 1 ? 21 : -1);    // Y = C W  (overwrites V)
            if (it == 1) SSTAMP(9);
        }
        spk_ritz_orth(V, R, sh, it == 2 ? 40 : -1);
        if (it == 1) SSTAMP(10);
        top4 = sh.top4;
        if (spk_converged(top4, it, prev_sum, prev_delta)) {
            conv = 1;
            break;
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
