// Small dense building blocks shared by the eigen kernels (eigen.hip) and the sparse route (sparse.hip):
// 16-wide block operations on an R x NB block held in LDS, all on v_mfma_f64_16x16x4_f64, plus the one-wave
// Jacobi eigen-solver for the 16 x 16 (or 8 x 8) projected matrices.  NB is the number of live columns
// (16 or 8; with 8 the MFMA tiles are half empty, which is irrelevant at these sizes), VP the row pitch of
// the block in doubles (odd, so that tile-wise and row-wise LDS reads are conflict-free).
#pragma once
#include "common.h"

#ifndef EIG_STAMP
#define EIG_STAMP(i)   // cycle stamps of the diagnostic build (eigen.hip, -DEIG_STAMPS)
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));

#define EIG_THREADS 512
#define EIG_WAVES 8
#define EIG_B 16
#define EIG_VP 17        // V row pitch in doubles (odd pitch: row-wise and tile-wise reads conflict-free)
#define EIG_MAXT 8       // row tiles of 16 per wave -> R_pad <= 8 * 8 * 16 = 1024
#define EIG_MAXIT 400

struct EigShared {
    double H[EIG_B * EIG_VP];
    double Q[EIG_B * EIG_VP];
    double G1[EIG_B * EIG_VP];  // V^T G V of the current block (first power of G: the values the score is taken from)
    double T[EIG_B * EIG_VP];   // L^-T (upper triangular), dead columns zeroed
    double top4;
    double sum_all, theta4;   // ritz_orth_nb: sum of all NB Ritz values and the 4th largest (certified stop of the dense route)
    double part[(EIG_WAVES / 2) * 256];  // cross-wave reduction buffer (two waves share a slot)
    double red[EIG_WAVES];
    double theta[EIG_B];
    int dead[EIG_B];
    int flag;
};

__device__ __forceinline__ double hash_unit(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (double)x * (2.0 / 4294967296.0) - 1.0;
}

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Sum one 16 x 16 MFMA accumulator over the 8 waves of the block into out (16 x EIG_VP, LDS).
// Waves 4-7 deposit first, waves 0-3 add theirs on top, then 256 threads add the 4 slots in a
// fixed order (deterministic).  Ends with a barrier.
__device__ __forceinline__ void reduce16(const double4_t& acc, EigShared& sh, double* out) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    if (w >= EIG_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sh.part[(w - EIG_WAVES / 2) * 256 + (fk + 4 * r) * 16 + fr] = acc[r];
    }
    __syncthreads();
    if (w < EIG_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sh.part[w * 256 + (fk + 4 * r) * 16 + fr] += acc[r];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < EIG_WAVES / 2; ++i) s += sh.part[i * 256 + threadIdx.x];
        out[(threadIdx.x >> 4) * EIG_VP + (threadIdx.x & 15)] = s;
    }
    __syncthreads();
}

// S = X^T X for the R x 16 array X in LDS (rows >= Rp are not touched): per-wave MFMA partials
// into sh.part, then summed into `out` (16 x EIG_VP).  Ends with a barrier.
template <int NB, int VP>
__device__ __forceinline__ void gram_nb(const double* X, int Rp, EigShared& sh, double* out) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    double4_t acc = {0, 0, 0, 0};
    for (int r0 = w * 4; r0 < Rp; r0 += EIG_WAVES * 4) {
        const double x = (NB == 16 || fr < NB) ? X[(r0 + fk) * VP + fr] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    reduce16(acc, sh, out);
}

// X <- X * B for the R x 16 array X in LDS and a 16 x 16 matrix B in LDS, by MFMA, in place
// (each wave owns whole 16-row tiles: all reads of a tile precede its writes).  Ends with a barrier.
template <int NB, int VP>
__device__ __forceinline__ void rowmul_nb(double* X, int Rp, const double* B) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int KB = NB / 4;
    double bfrag[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) bfrag[kb] = B[(kb * 4 + fk) * EIG_VP + fr];
    for (int tile = w; tile < (Rp >> 4); tile += EIG_WAVES) {
        double4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const double a = X[(tile * 16 + fr) * VP + kb * 4 + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[kb], acc, 0, 0, 0);
        }
        if (NB == 16 || fr < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(tile * 16 + fk + 4 * r) * VP + fr] = acc[r];
        }
    }
    __syncthreads();
}


// Accurate 1/sqrt(x) from an f32 seed and two Newton steps (fp64 sqrt/div sequences are serial
// bottlenecks in a one-wave Jacobi).
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = (double)__frsqrt_rn((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Jacobi rotation (c, s) for the symmetric 2 x 2 [app apq; apq aqq].  The angle is evaluated in f32
// (cheap), c = rsqrt(1 + t^2) and s = t c in fp64, so the rotation is orthogonal to fp64 accuracy for
// ANY t - an inexact angle only leaves a residual ~1e-7 |apq| that the next sweep removes.
// One wave runs the whole Jacobi with nothing to overlap its latencies, so this is a dependent chain that is paid
// 15 times a sweep: tau comes from the hardware fp64 reciprocal (the quotient of two fp64 numbers needs no scaling
// into the f32 range first - the round-1 form spent most of its time in ilogb / scalbn), the f32 part uses the
// hardware reciprocal and square root.
// (branch-free: a lane needs two rotations per round and the compiler interleaves the two dependent chains only when
// they are straight-line code; with the `if`s of the round-1 form they ran one after the other)
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
    // A coupling below 1e-15 of its two diagonal entries is rounding noise and moves no eigenvalue (second order: 1e-30),
    // but between two (nearly) EQUAL diagonal entries it would be 'removed' by a 45 degree rotation: the orthonormalised
    // strong columns of ritz_orth_nb's second pass (Gram block = I + noise) were scrambled that way every round, the
    // next round's projected matrix was full again and its Jacobi needed 6 sweeps instead of starting near-diagonal.
    const bool rot = apq * apq > 1e-30 * fabs(app * aqq) && apq != 0.0;
    const double d = aqq - app;
    const double tau = d * __builtin_amdgcn_rcp(2.0 * apq);
    // general angle in f32 (|tau| <= 1e18 keeps tau^2 inside the f32 range; an inf / nan tau takes the other form)
    const float tf = (float)tau;
    const float r = __builtin_amdgcn_rcpf(fabsf(tf) + __builtin_amdgcn_sqrtf(1.0f + tf * tf));
    const double t_gen = (double)copysignf(r, tf);
    // tiny angle t = 1 / (2 tau) = apq / d; apq so small that its reciprocal is inf, or d = 0 next to it: no rotation
    double t_tiny = apq * __builtin_amdgcn_rcp(d);
    t_tiny = fabs(t_tiny) <= 1e-17 ? t_tiny : 0.0;
    double t = fabs(tau) <= 1e18 ? t_gen : t_tiny;
    t = rot ? t : 0.0;
    c = rsqrt_nr(1.0 + t * t);   // exactly 1 for t = 0
    s = t * c;
}

// Eigen-decomposition of the symmetric positive semi-definite 16 x 16 matrix sh.H by parallel-order
// (round-robin) Jacobi, wave 0 only: eigenvalues to sh.theta, eigenvectors to the columns of sh.Q.
// Lane (a, b) owns the 2 x 2 block {p_a, q_a} x {p_b, q_b} of H for the round's 8 disjoint pairs and
// computes new block = J_a^T block J_b in registers; it derives BOTH rotations itself from the two
// diagonal blocks (extra broadcast LDS reads instead of cross-lane shuffles), so a round is one LDS
// round trip, ~100 flops and one wave-level sync.  Convergence is judged relatively (|h_ij|^2 against
// h_ii h_jj), which is what gives Jacobi its high relative accuracy on PSD matrices.  Ends with a barrier.
template <int NB>
__device__ __forceinline__ void jacobi_core(double* H, double* Q, double* theta, bool diag) {
    {
        const int lane = threadIdx.x & 63;
        constexpr int NP = NB / 2;  // disjoint pairs per round; lanes 0 .. NP*NP-1 each own a 2 x 2 block
        const bool act = lane < NP * NP;
        const int pa = act ? lane / NP : 0, pb = act ? lane % NP : 0;
        for (int e = lane; e < EIG_B * EIG_B; e += 64) Q[(e >> 4) * EIG_VP + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        wave_sync_lds();
#pragma unroll 1
        for (int sweep = 0; sweep < 15; ++sweep) {
            double dmx = lane < NB ? fabs(H[lane * EIG_VP + lane]) : 0.0;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) dmx = fmax(dmx, __shfl_xor(dmx, d, 64));
            dmx = __shfl(dmx, 0, 64);
            bool open = false;   // some coupling is still above 1e-11 of its two diagonal entries (v^2 > 1e-22 h_ii h_jj)
            for (int e = lane; e < EIG_B * EIG_B; e += 64) {
                const int i = e >> 4, j = e & 15;
                if (i < j && j < NB) {
                    const double v = H[i * EIG_VP + j];
                    const double dd = fabs(H[i * EIG_VP + i] * H[j * EIG_VP + j]);
                    // couplings below 1e-20 of the largest eigenvalue cannot matter (directions that small are
                    // noise or dead) and would otherwise keep the sweeps going on rounding residue
                    open = open || (v * v > 1e-40 * dmx * dmx && !(v * v <= 1e-22 * dd));
                }
            }
            const double rel = __builtin_amdgcn_ballot_w64(open) != 0ull ? 1.0 : 0.0;
#ifdef EIG_STAMPS
            if (diag && blockIdx.x == 0 && g_eig_stamp_on && g_eig_stamps[39] == 0 && sweep < 10) {
                double mx = 0; int nopen = 0;
                for (int e = lane; e < EIG_B * EIG_B; e += 64) {
                    const int i = e >> 4, j = e & 15;
                    if (i < j && j < NB) {
                        const double v = H[i * EIG_VP + j];
                        const double dd = fabs(H[i * EIG_VP + i] * H[j * EIG_VP + j]);
                        mx = fmax(mx, fabs(v) / dmx);
                        nopen += (v * v > 1e-40 * dmx * dmx && !(v * v <= 1e-22 * dd)) ? 1 : 0;
                    }
                }
                for (int d = 32; d >= 1; d >>= 1) { mx = fmax(mx, __shfl_xor(mx, d, 64)); nopen += __shfl_xor(nopen, d, 64); }
                if (lane == 0) g_eig_stamps[24 + sweep] = (long long)(-log10(mx + 1e-300) * 10) * 1000 + nopen;
                if (lane < 16 && sweep == 0) g_eig_stamps[56 + (lane & 7)] = 0;
            }
#endif
#ifdef EIG_STAMPS
            if (diag && lane == 0 && blockIdx.x == 0 && g_eig_stamp_on) {
                const int slot = g_eig_stamps[39] < 8 ? (int)g_eig_stamps[39] : 7;   // [39] = Jacobi calls so far
                g_eig_stamps[40 + slot] = sweep;
                g_eig_stamps[48 + slot] = (long long)(rel * 1e30 < 9e18 ? rel * 1e30 : 9e18);
            }
#endif
            if (!(rel > 1e-22)) break;
#ifndef EIG_JACOBI_UNROLL
#pragma unroll 1
#endif
            for (int round = 0; round < NB - 1; ++round) {
                int ip, iq, jp, jq;  // rows of pair a, cols of pair b
                // (round + pa and round + NB - 1 - pa are below 2 (NB - 1): one conditional subtraction, no division)
                const int a1 = round + pa, a2 = round + (NB - 1) - pa, b1 = round + pb, b2 = round + (NB - 1) - pb;
                ip = pa == 0 ? NB - 1 : (a1 >= NB - 1 ? a1 - (NB - 1) : a1);
                iq = pa == 0 ? round : (a2 >= NB - 1 ? a2 - (NB - 1) : a2);
                if (ip > iq) { const int t = ip; ip = iq; iq = t; }
                jp = pb == 0 ? NB - 1 : (b1 >= NB - 1 ? b1 - (NB - 1) : b1);
                jq = pb == 0 ? round : (b2 >= NB - 1 ? b2 - (NB - 1) : b2);
                if (jp > jq) { const int t = jp; jp = jq; jq = t; }
                const double hpp = H[ip * EIG_VP + jp], hpq = H[ip * EIG_VP + jq];
                const double hqp = H[iq * EIG_VP + jp], hqq = H[iq * EIG_VP + jq];
                const double a_pp = H[ip * EIG_VP + ip], a_qq = H[iq * EIG_VP + iq], a_pq = H[ip * EIG_VP + iq];
                const double b_pp = H[jp * EIG_VP + jp], b_qq = H[jq * EIG_VP + jq], b_pq = H[jp * EIG_VP + jq];
                const double q0p = Q[(2 * pa) * EIG_VP + jp], q0q = Q[(2 * pa) * EIG_VP + jq];
                const double q1p = Q[(2 * pa + 1) * EIG_VP + jp], q1q = Q[(2 * pa + 1) * EIG_VP + jq];
                double ca, sa, cb, sb;
                jacobi_cs(a_pp, a_qq, a_pq, ca, sa);
                jacobi_cs(b_pp, b_qq, b_pq, cb, sb);
                // rows: [p'; q'] = [c -s; s c] [p; q]   (J^T from the left, J = [c s; -s c])
                const double rpp = ca * hpp - sa * hqp, rpq = ca * hpq - sa * hqq;
                const double rqp = sa * hpp + ca * hqp, rqq = sa * hpq + ca * hqq;
                wave_sync_lds();  // every lane has read before anyone writes
                // cols: [p' q'] = [p q] [c s; -s c]
                if (act) {
                H[ip * EIG_VP + jp] = cb * rpp - sb * rpq;
                H[ip * EIG_VP + jq] = sb * rpp + cb * rpq;
                H[iq * EIG_VP + jp] = cb * rqp - sb * rqq;
                H[iq * EIG_VP + jq] = sb * rqp + cb * rqq;
                // Q <- Q J: lane (a, b) updates rows {2a, 2a+1} of the column pair b
                Q[(2 * pa) * EIG_VP + jp] = cb * q0p - sb * q0q;
                Q[(2 * pa) * EIG_VP + jq] = sb * q0p + cb * q0q;
                Q[(2 * pa + 1) * EIG_VP + jp] = cb * q1p - sb * q1q;
                Q[(2 * pa + 1) * EIG_VP + jq] = sb * q1p + cb * q1q;
                }
                wave_sync_lds();
            }
        }
        if (lane < EIG_B) theta[lane] = lane < NB ? H[lane * EIG_VP + lane] : 0.0;
#ifdef EIG_STAMPS
        if (diag && lane == 0 && blockIdx.x == 0 && g_eig_stamp_on) g_eig_stamps[39] += 1;
#endif
    }
}

// Wave 0 runs the Jacobi on sh.H / sh.Q / sh.theta.  Ends with a barrier.
template <int NB>
__device__ __forceinline__ void jacobi_nb(EigShared& sh) {
    if (threadIdx.x < 64) jacobi_core<NB>(sh.H, sh.Q, sh.theta, true);
    __syncthreads();
}

// Second 16 x 16 problem solved NEXT TO the first one by wave 1 (eigen.hip: the first-power projection G1 of the dense
// route, whose eigenvalues the acceptance test wants in the round that stops - one wave's Jacobi is 60 - 90 us of pure
// latency and the other 7 waves of the workgroup are idle meanwhile).  top4 = sum of the four largest eigenvalues of H.
struct EigSpec {
    double H[EIG_B * EIG_VP];
    double Q[EIG_B * EIG_VP];
    double theta[EIG_B];
    double top4;
};
template <int NB>
__device__ __forceinline__ void jacobi_dual(EigShared& sh, EigSpec& sp2) {
    const int w = sp_wave_id();
    if (w == 0) jacobi_core<NB>(sh.H, sh.Q, sh.theta, true);
    else if (w == 1) {
        jacobi_core<EIG_B>(sp2.H, sp2.Q, sp2.theta, false);
        const int lane = threadIdx.x & 63;
        const double th = lane < EIG_B ? sp2.theta[lane] : -1e300;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < EIG_B; ++j) {
            const double o = __shfl(th, j, 64);
            rank += (o > th || (o == th && j < lane)) ? 1 : 0;
        }
        double pick = (lane < EIG_B && rank < 4) ? fmax(th, 0.0) : 0.0;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) pick += __shfl_xor(pick, d, 64);
        if (lane == 0) sp2.top4 = pick;
    }
    __syncthreads();
}

// One Rayleigh-Ritz + orthonormalisation step on the R x 16 block X = G V (LDS), in place:
//     S = X^T X = V^T G^2 V            (MFMA)
//     S = P D P^T                      (one 16 x 16 Jacobi)
//     X <- X P D^-1/2                  (MFMA; columns = orthonormal Ritz vectors of G^2 on span(V), images under G)
//     polish: X <- X (1.5 I - 0.5 X^T X) until ||X^T X - I||_max <= 2e-15   (Newton-Schulz, MFMA)
// sqrt(D_i) are Ritz values of G (from G^2 on the same subspace: lower bounds, second-order accurate like
// the ones of V^T G V); sh.top4 = sum of the four largest.  Directions with D_i <= 3e-15 D_max are dead
// (zero columns: exactly singular input, or R < 16).  No Cholesky, no serial 16-step chains, and no
// data-dependent fallback path.  When `values_only` the block is left untouched after the Jacobi.
template <int NB, int VP>
__device__ __forceinline__ void polish_nb(double* X, int Rp, EigShared& sh);

template <int NB, int VP>
__device__ __forceinline__ void ritz_orth_nb(double* X, int Rp, EigShared& sh, EigSpec* spec = nullptr) {
    gram_nb<NB, VP>(X, Rp, sh, sh.H);
    EIG_STAMP(10);
#ifdef EIG_STAMPS
    if (blockIdx.x == 0 && g_eig_stamp_on && threadIdx.x < 256) g_eig_dump[threadIdx.x] = sh.H[(threadIdx.x >> 4) * EIG_VP + (threadIdx.x & 15)];
    __syncthreads();
#endif
    if (spec) jacobi_dual<NB>(sh, *spec);   // (uniform)
    else jacobi_nb<NB>(sh);
    EIG_STAMP(11);
    bool weak;
    // top-4 sum + T = P D^-1/2 (256 threads)
    {
        double dmax = 0;
        for (int k = 0; k < EIG_B; ++k) dmax = fmax(dmax, sh.theta[k]);
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const double th = lane < EIG_B ? sqrt(fmax(sh.theta[lane], 0.0)) : -1.0;
            int rank = 0;
#pragma unroll
            for (int j = 0; j < EIG_B; ++j) {
                const double o = __shfl(th, j, 64);
                rank += (o > th || (o == th && j < lane)) ? 1 : 0;
            }
            double pick = (lane < EIG_B && rank < 4) ? th : 0.0;
            double all = lane < EIG_B ? fmax(th, 0.0) : 0.0;
            double fourth = (lane < EIG_B && rank == 3) ? th : 0.0;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                pick += __shfl_xor(pick, d, 64);
                all += __shfl_xor(all, d, 64);
                fourth += __shfl_xor(fourth, d, 64);
            }
            if (lane == 0) {
                sh.top4 = pick;
                sh.sum_all = all;
                sh.theta4 = fourth;
            }
        }
        // S = Y^T Y carries absolute rounding errors of ~1e-16 dmax, so an eigenvalue below ~3e-15 dmax is noise, and
        // normalising a column by the square root of noise made X^T X explode in the polish below (nan scores on a
        // probability-scaled table - found by the randomised tests).  The COLUMN Y q_j itself is fine, though: it is a
        // sum of O(|Y|) terms, good to ~1e-16 lambda_1 absolute, i.e. it still carries the direction of an eigenvalue
        // lambda_j ~ 1e-9 lambda_1 of G with 7 digits - and on a matrix of numerical rank < 4 such a direction decides
        // the score.  So a weak column is scaled by the clamped factor only (its norm stays < 1) and, when there are
        // weak columns, the block is orthonormalised a second time from ITS OWN Gram matrix, which sees them at their
        // own scale (CholeskyQR2 in Jacobi form).  Directions that are still < 1e-6 after the clamped scaling are dead.
        weak = false;
        for (int k = 0; k < NB; ++k) weak = weak || !(sh.theta[k] > 1e-10 * dmax);
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            const double dj = fmax(sh.theta[j], 3e-15 * dmax);
            // (the f32 seed of rsqrt_nr needs its argument inside the f32 range: G^2 values of a matrix scaled by 1e-8 or
            // 1e+5 are not - 0 or inf came back and the score was 1 - so the exponent is split off first, exactly)
            const int ej = (dj > 0) ? (ilogb(dj) & ~1) : 0;
            const double rj = dj > 0 ? scalbn(rsqrt_nr(scalbn(dj, -ej)), -(ej / 2)) : 0.0;
            sh.T[i * EIG_VP + j] = (i < NB && j < NB) ? sh.Q[i * EIG_VP + j] * rj : 0.0;
        }
        __syncthreads();
    }
    rowmul_nb<NB, VP>(X, Rp, sh.T);
    EIG_STAMP(12);
    if (weak) {
        gram_nb<NB, VP>(X, Rp, sh, sh.H);
        jacobi_nb<NB>(sh);
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            const double dj = sh.theta[j];
            const double rj = dj > 1e-12 ? 1.0 / sqrt(dj) : 0.0;
            sh.T[i * EIG_VP + j] = (i < NB && j < NB) ? sh.Q[i * EIG_VP + j] * rj : 0.0;
        }
        __syncthreads();
        rowmul_nb<NB, VP>(X, Rp, sh.T);
    }
    polish_nb<NB, VP>(X, Rp, sh);
    EIG_STAMP(23);
}

// Newton-Schulz polish of a nearly orthonormal R x NB block in LDS: X <- X (1.5 I - 0.5 X^T X) until
// ||X^T X - I||_max <= 2e-15 (columns whose squared norm is below 0.25 are dead and stay as they are).  Converges
// quadratically from any block with ||X^T X - I||_2 < 1.  Ends with a barrier.
template <int NB, int VP>
__device__ __forceinline__ void polish_nb(double* X, int Rp, EigShared& sh) {
    for (int iter = 0; iter < 10; ++iter) {
        gram_nb<NB, VP>(X, Rp, sh, sh.H);
        double err = 0;
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            const double sv = sh.H[i * EIG_VP + j];
            const double target = (i == j && sh.H[i * EIG_VP + i] > 0.25) ? 1.0 : 0.0;  // dead columns stay 0
            err = (i < NB && j < NB) ? fabs(sv - target) : 0.0;
            sh.T[i * EIG_VP + j] = (i < NB && j < NB) ? (i == j ? 1.5 : 0.0) - 0.5 * sv : 0.0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) err = fmax(err, __shfl_xor(err, d, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh.red[sp_wave_id()] = err;
        __syncthreads();
        double emx = 0;
        for (int k = 0; k < EIG_WAVES; ++k) emx = fmax(emx, sh.red[k]);
        __syncthreads();
        EIG_STAMP(13 + iter);
        if (emx <= 2e-15) break;
        rowmul_nb<NB, VP>(X, Rp, sh.T);
    }
}

// Orthonormalise a block whose columns are already close to orthogonal (the start block: unit vectors + 2 % noise):
// columns scaled to unit length, then the polish.  No eigen-decomposition - the span is what matters to the iteration,
// and a 16 x 16 Jacobi from a full matrix is 9 sweeps of one wave (~190 k cycles, the bulk of k_eig_init in round 1).
template <int NB, int VP>
__device__ __forceinline__ void orth_near_nb(double* X, int Rp, EigShared& sh) {
    gram_nb<NB, VP>(X, Rp, sh, sh.H);
    if (threadIdx.x < 256) {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        const double dj = sh.H[j * EIG_VP + j];
        sh.T[i * EIG_VP + j] = (i == j && i < NB && dj > 0) ? rsqrt_nr(dj) : 0.0;
    }
    __syncthreads();
    rowmul_nb<NB, VP>(X, Rp, sh.T);
    polish_nb<NB, VP>(X, Rp, sh);
}

// First-power Rayleigh-Ritz for the final score.  The iteration itself works on G^2 (ritz_orth_nb): its Ritz values
// sqrt(D_i) carry the rounding of S = Y^T Y, i.e. ~1e-16 lambda_1^2 absolute, which is nothing for a well separated
// spectrum but ruins an eigenvalue lambda_4 ~ 1e-7 lambda_1 (scores ~1e-4 off by 5e-5 on nearly rank-4 flattenings -
// found by the randomised tests).  The same subspace span(V) also gives G1 = V^T (G V) = V^T Y at the cost of one more
// R x 16 x 16 MFMA product, and its eigenvalues are Ritz values of G with absolute error ~1e-16 lambda_1.
// proj_first_power: Y in LDS (pitch VP), V^T in global memory (16 x vp, store_vt layout) -> sh.G1.  Ends with a barrier.
template <int VP>
__device__ __forceinline__ void proj_first_power(const double* Y, int Rp, const double* __restrict__ Vt, int vp,
                                                 EigShared& sh) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int fr = lane & 15, fk = lane >> 4;
    double4_t acc = {0, 0, 0, 0};
    // 16 rows a step: a lane loads 4 consecutive entries of its V^T row (32 B; the 4 lanes of a row read one 128-B
    // segment) and issues 4 MFMAs, MFMA j summing the rows {r0 + 4 g + j : g = 0..3} - the access rule of k_eig_gv.
    // (One 8-byte load per MFMA, 16 rows x 4 scattered doubles per wave-instruction, cost 24 k cycles on a 1024-row side.)
    typedef double f64x2_t __attribute__((ext_vector_type(2)));
#pragma unroll 4
    for (int r0 = w * 16; r0 < Rp; r0 += EIG_WAVES * 16) {
        const double* __restrict__ p = Vt + (int64_t)fr * vp + r0 + 4 * fk;
        const f64x2_t v01 = *reinterpret_cast<const f64x2_t*>(p), v23 = *reinterpret_cast<const f64x2_t*>(p + 2);
        const double* y = Y + (r0 + 4 * fk) * VP + fr;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v01.x, y[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v01.y, y[VP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v23.x, y[2 * VP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v23.y, y[3 * VP], acc, 0, 0, 0);
    }
    reduce16(acc, sh, sh.G1);
}

// sh.top4 = sum of the four largest eigenvalues of the symmetric 16 x 16 matrix in sh.H (destroyed; sh.Q too).
// Ends with a barrier.
__device__ __forceinline__ void top4_of_H(EigShared& sh) {
    jacobi_nb<EIG_B>(sh);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const double th = lane < EIG_B ? sh.theta[lane] : -1e300;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < EIG_B; ++j) {
            const double o = __shfl(th, j, 64);
            rank += (o > th || (o == th && j < lane)) ? 1 : 0;
        }
        double pick = (lane < EIG_B && rank < 4) ? fmax(th, 0.0) : 0.0;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) pick += __shfl_xor(pick, d, 64);
        if (lane == 0) sh.top4 = pick;
    }
    __syncthreads();
}

// sh.top4 = sum of the four largest eigenvalues of the (symmetrised) sh.G1.
__device__ __forceinline__ void first_power_top4(EigShared& sh) {
    if (threadIdx.x < 256) {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        sh.H[i * EIG_VP + j] = 0.5 * (sh.G1[i * EIG_VP + j] + sh.G1[j * EIG_VP + i]);
    }
    __syncthreads();
    top4_of_H(sh);
}

struct EigState {
    double trace, prev_sum, prev_delta, prev_ratio, top4;
    double f_prev_sum, f_prev_delta, f_prev_ratio;   // the same bookkeeping on the first-power sums (fp_it > 0)
    int it, done, R, fp_it;
};

#define EIG_NFAST 5

__device__ __forceinline__ double block_sum(double v, EigShared& sh) {
    const int lane = threadIdx.x & 63, w = sp_wave_id();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane == 0) sh.red[w] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < EIG_WAVES; ++i) t += sh.red[i];
    return t;
}

// Convergence bookkeeping shared by k_eig_rr and k_eig_finish (uniform across the block).
// The tail is priced with the LARGER of the last two error ratios: successive ratios of a sum of decaying components
// can only grow, and a single early ratio stopped some splits with 1e-8 left in the score (randomised tests).
__device__ __forceinline__ bool update_convergence(double s4, int it, double& prev_sum, double& prev_delta,
                                                   double& prev_ratio) {
    bool conv = false;
    const double delta = fabs(s4 - prev_sum);
    double ratio = 1.0;
    if (it >= 2) {
        ratio = prev_delta > 0 ? delta / prev_delta : 0.0;
        ratio = fmin(fmax(ratio, 0.0), 0.9999);
        const double r = fmax(ratio, prev_ratio);
        const double tail = delta * r / (1.0 - r);
        if (delta <= 4e-16 * s4 || (it >= 4 && tail <= 1e-14 * s4)) conv = true;
    }
    prev_ratio = ratio;
    prev_delta = delta;
    prev_sum = s4;
    return conv;
}

// Certified stop (round 2; the sparse route's rule (c), DESIGN 3a, carried over to the 16-wide block).  Everything outside
// the block weighs rest = trace - (sum of all 16 Ritz values) >= lambda_17, the 4th Ritz value theta_4 <= lambda_4, so no
// component of the top-4 sum's error decays slower than q = (rest / theta_4)^2 per product (Ritz values are second order
// in the angle) and what is still missing after a step of size delta is at most delta q / (1 - q).  Phylogenetic
// flattenings have rest / theta_4 ~ 1e-4 ... 2e-2: the bound certifies the sum after the 3rd product, where the two-ratio
// rule of update_convergence cannot speak before the 4th - one G V product and one Rayleigh-Ritz round less.
__device__ __forceinline__ bool certified_stop(double s4, double delta, int it, double trace, const EigShared& sh) {
    if (it < 3) return false;
    const double rest = trace - sh.sum_all;
    if (!(rest >= 0.0) || !(sh.theta4 > 0.0)) return false;
    const double rho = rest / sh.theta4;
    if (!(rho < 0.25)) return false;
    const double q = rho * rho;
    // Tolerance on the sum as in the sparse route (spk_converged): the score is sqrt(1 - s / trace), so an error e of the
    // sum moves it by e / (2 score trace); 4e-11 score trace keeps that below 2e-11, capped at 1e-12 of the sum and never
    // asked below the sum's rounding floor.  The bound is conservative - rest overestimates lambda_17 ~30x on real
    // alignments: where it says 1e-12 the sum is typically good to 1e-15 - so a fixed 1e-15 certified only a quarter of
    // config 2's splits after the 3rd product; this certifies all of them, and the 4th round of launches is empty.
    const double sx = sqrt(fmax(trace - s4, 0.0) * trace);   // = score * trace
    const double tol = fmax(fmin(1e-12 * s4, 4e-11 * sx), 4e-15 * s4);
    return delta * q / (1.0 - q) <= tol;
}

// Final acceptance, shared by k_eig_rr and k_eig_finish (uniform).  `g2_conv`: the G^2 sums have settled this round.
// The first-power sum f of the same subspace (sh.G1) is then evaluated: if it agrees with the G^2 sum to 1e-13 the G^2
// iteration resolved all four values and f is final.  If not, the 4th value is below what G^2 can see (a matrix of
// numerical rank < 4: lambda_4 inside the cluster of its noise eigenvalues) and the G^2 sums are blind to its
// convergence: from then on f is evaluated every round and judged by the two-ratio rule itself.
__device__ __forceinline__ bool accept_first_power(bool g2_conv, double g2_sum, EigState& st, EigShared& sh,
                                                   const EigSpec* spec = nullptr) {
    // (also from the 6th product on: G^2 sums of a moderately ill-conditioned spectrum sit on their rounding floor, above
    // the tolerance, and would never settle)
    if (!g2_conv && st.fp_it == 0 && st.it < 6) return false;
    double f;
    if (spec) f = spec->top4;          // solved by wave 1 next to this round's G^2 Jacobi (jacobi_dual)
    else {
        first_power_top4(sh);
        f = sh.top4;
    }
    if (st.fp_it == 0 && g2_conv && fabs(f - g2_sum) <= 1e-13 * g2_sum) {
        st.top4 = f;
        return true;
    }
    st.fp_it += 1;
    const bool conv = update_convergence(f, st.fp_it, st.f_prev_sum, st.f_prev_delta, st.f_prev_ratio);
    st.top4 = f;
    return conv;
}

__device__ __forceinline__ void write_score(double top4, double tr, int it, bool converged, double* scores,
                                            int* status, int sid) {
    const double op = 1.0 - top4 / tr;
    scores[sid] = sqrt(op > 0 ? op : 0.0);
    status[sid] = (converged ? 0 : 1) | (it << 8);
}

// V (LDS, R x 16, pitch EIG_VP) <-> Vt (global, 16 x vp column-major: the B operand of k_eig_gv reads
// 4 consecutive k per lane)
__device__ __forceinline__ void store_vt(const double* V, int Rp, double* __restrict__ Vt, int vp) {
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) {
        const int col = e / Rp, row = e % Rp;
        Vt[(int64_t)col * vp + row] = V[row * EIG_VP + col];
    }
}
__device__ __forceinline__ void load_vt(double* V, int Rp, const double* __restrict__ Vt, int vp) {
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) {
        const int col = e / Rp, row = e % Rp;
        V[row * EIG_VP + col] = Vt[(int64_t)col * vp + row];
    }
}

