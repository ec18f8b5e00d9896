// Top-4 eigenvalues of the Gram matrix + the split score.
//
// Second half of the replacement for splitp/phylogenetics.py:280-300 (__dense_split_score):
//     score = (1 - sum(sigma[:4]^2) / sum(sigma^2)) ** 0.5
// With G = C C^T:  sum(sigma^2) = trace(G) (exact for integer counts) and sigma[:4]^2 are the four
// largest eigenvalues of G.  They are found by block orthogonal (subspace) iteration with a
// Rayleigh-Ritz step, block width 16 (the N of v_mfma_f64_16x16x4_f64):
//     Y = G V            fp64 MFMA, G streamed from L2/HBM (symmetric: read as columns, 128-B segments),
//                        V (R x 16) resident in LDS, Y accumulators in registers
//     H = V^T Y          fp64 MFMA straight from the Y accumulators (C/D registers are B operands)
//     H = Q Theta Q^T    16 x 16 parallel-order Jacobi by one wave
//     Z = Y Q            (= G times the Ritz vectors; columns nearly orthogonal)
//     V = orth(Z)        Cholesky-QR twice (Gram by MFMA, 16 x 16 Cholesky, row-wise solve)
// until the sum of the four largest Ritz values stops moving (geometric-tail estimate below
// 1e-14 relative).  Phylogenetic flattenings have lambda_17 / lambda_4 < 1e-3, so this takes 3-4
// products.
//
// Launch structure: the product Y = G V is the only heavy step and G (up to 1024^2 doubles per
// split) has to stream from HBM once per product, so it runs as its own kernel over ALL splits
// with one workgroup per 64 rows (k_eig_gv: the whole chip pulls on HBM), while the small per-split
// algebra (H, Jacobi, Cholesky-QR on an R x 16 block held in LDS) runs one workgroup per split
// (k_eig_init, k_eig_rr).  EIG_NFAST product rounds are enqueued without host synchronisation;
// splits that have converged turn their workgroups into no-ops.  k_eig_finish then takes any
// split that is still not converged (arbitrary matrices without a spectral gap) to convergence
// inside one workgroup, capped at EIG_MAXIT (status bit 0).
#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define EIG_THREADS 512
#define EIG_WAVES 8
#define EIG_B 16
#define EIG_VP 17        // V row pitch in doubles (odd pitch: row-wise and tile-wise reads conflict-free)
#define EIG_MAXT 8       // row tiles of 16 per wave -> R_pad <= 8 * 8 * 16 = 1024
#define EIG_MAXR 1024
#define EIG_MAXIT 400

struct EigShared {
    double H[EIG_B * EIG_VP];
    double Q[EIG_B * EIG_VP];
    double L[EIG_B * EIG_VP];
    double T[EIG_B * EIG_VP];   // L^-T (upper triangular), dead columns zeroed
    double top4;
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
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
__device__ __forceinline__ void gram16(const double* X, int Rp, EigShared& sh, double* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    double4_t acc = {0, 0, 0, 0};
    for (int r0 = w * 4; r0 < Rp; r0 += EIG_WAVES * 4) {
        const double x = X[(r0 + fk) * EIG_VP + fr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    reduce16(acc, sh, out);
}

// X <- X * B for the R x 16 array X in LDS and a 16 x 16 matrix B in LDS, by MFMA, in place
// (each wave owns whole 16-row tiles: all reads of a tile precede its writes).  Ends with a barrier.
__device__ __forceinline__ void rowmul16(double* X, int Rp, const double* B) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    double bfrag[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bfrag[kb] = B[(kb * 4 + fk) * EIG_VP + fr];
    for (int tile = w; tile < (Rp >> 4); tile += EIG_WAVES) {
        double4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double a = X[(tile * 16 + fr) * EIG_VP + kb * 4 + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[kb], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(tile * 16 + fk + 4 * r) * EIG_VP + fr] = acc[r];
    }
    __syncthreads();
}

#ifdef EIG_STAMPS
__device__ long long g_eig_stamps[64];
#define STAMP(i)                                                                         \
    do {                                                                                 \
        __syncthreads();                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_eig_stamps[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(i)
#endif

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
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
    c = 1.0;
    s = 0.0;
    if (apq * apq > 1e-40 * fabs(app * aqq) && apq != 0.0) {
        const float num = (float)(aqq - app), den = 2.0f * (float)apq;
        float tf;
        if (fabsf(num) > 1e18f * fabsf(den)) {
            tf = den / (2.0f * num);  // tiny angle; avoids inf / nan in the f32 quotient
        } else {
            const float tau = num / den;
            tf = (tau >= 0.f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
        }
        const double t = (double)tf;
        c = rsqrt_nr(1.0 + t * t);
        s = t * c;
    }
}

// Eigen-decomposition of the symmetric positive semi-definite 16 x 16 matrix sh.H by parallel-order
// (round-robin) Jacobi, wave 0 only: eigenvalues to sh.theta, eigenvectors to the columns of sh.Q.
// Lane (a, b) owns the 2 x 2 block {p_a, q_a} x {p_b, q_b} of H for the round's 8 disjoint pairs and
// computes new block = J_a^T block J_b in registers; it derives BOTH rotations itself from the two
// diagonal blocks (extra broadcast LDS reads instead of cross-lane shuffles), so a round is one LDS
// round trip, ~100 flops and one wave-level sync.  Convergence is judged relatively (|h_ij|^2 against
// h_ii h_jj), which is what gives Jacobi its high relative accuracy on PSD matrices.  Ends with a barrier.
__device__ __forceinline__ void jacobi16(EigShared& sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int pa = lane >> 3, pb = lane & 7;
        for (int e = lane; e < EIG_B * EIG_B; e += 64) sh.Q[(e >> 4) * EIG_VP + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        wave_sync_lds();
        int sweep_count = 0;
        for (int sweep = 0; sweep < 15; ++sweep) {
            sweep_count = sweep;
            double rel = 0, dmx = 0;
            for (int k = 0; k < EIG_B; ++k) dmx = fmax(dmx, fabs(sh.H[k * EIG_VP + k]));
            for (int e = lane; e < EIG_B * EIG_B; e += 64) {
                const int i = e >> 4, j = e & 15;
                if (i < j) {
                    const double v = sh.H[i * EIG_VP + j];
                    const double dd = fabs(sh.H[i * EIG_VP + i] * sh.H[j * EIG_VP + j]);
                    // couplings below 1e-20 of the largest eigenvalue cannot matter (directions that small are
                    // noise or dead) and would otherwise keep the sweeps going on rounding residue
                    if (v * v > 1e-40 * dmx * dmx) rel = fmax(rel, dd > 0 ? v * v / dd : 1.0);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) rel = fmax(rel, __shfl_xor(rel, d, 64));
            if (!(rel > 1e-22)) break;
            for (int round = 0; round < EIG_B - 1; ++round) {
                int ip, iq, jp, jq;  // rows of pair a, cols of pair b
                if (pa == 0) { ip = EIG_B - 1; iq = round; }
                else { ip = (round + pa) % (EIG_B - 1); iq = (round + (EIG_B - 1) - pa) % (EIG_B - 1); }
                if (ip > iq) { const int t = ip; ip = iq; iq = t; }
                if (pb == 0) { jp = EIG_B - 1; jq = round; }
                else { jp = (round + pb) % (EIG_B - 1); jq = (round + (EIG_B - 1) - pb) % (EIG_B - 1); }
                if (jp > jq) { const int t = jp; jp = jq; jq = t; }
                const double hpp = sh.H[ip * EIG_VP + jp], hpq = sh.H[ip * EIG_VP + jq];
                const double hqp = sh.H[iq * EIG_VP + jp], hqq = sh.H[iq * EIG_VP + jq];
                const double a_pp = sh.H[ip * EIG_VP + ip], a_qq = sh.H[iq * EIG_VP + iq], a_pq = sh.H[ip * EIG_VP + iq];
                const double b_pp = sh.H[jp * EIG_VP + jp], b_qq = sh.H[jq * EIG_VP + jq], b_pq = sh.H[jp * EIG_VP + jq];
                const double q0p = sh.Q[(2 * pa) * EIG_VP + jp], q0q = sh.Q[(2 * pa) * EIG_VP + jq];
                const double q1p = sh.Q[(2 * pa + 1) * EIG_VP + jp], q1q = sh.Q[(2 * pa + 1) * EIG_VP + jq];
                double ca, sa, cb, sb;
                jacobi_cs(a_pp, a_qq, a_pq, ca, sa);
                jacobi_cs(b_pp, b_qq, b_pq, cb, sb);
                // rows: [p'; q'] = [c -s; s c] [p; q]   (J^T from the left, J = [c s; -s c])
                const double rpp = ca * hpp - sa * hqp, rpq = ca * hpq - sa * hqq;
                const double rqp = sa * hpp + ca * hqp, rqq = sa * hpq + ca * hqq;
                wave_sync_lds();  // every lane has read before anyone writes
                // cols: [p' q'] = [p q] [c s; -s c]
                sh.H[ip * EIG_VP + jp] = cb * rpp - sb * rpq;
                sh.H[ip * EIG_VP + jq] = sb * rpp + cb * rpq;
                sh.H[iq * EIG_VP + jp] = cb * rqp - sb * rqq;
                sh.H[iq * EIG_VP + jq] = sb * rqp + cb * rqq;
                // Q <- Q J: lane (a, b) updates rows {2a, 2a+1} of the column pair b
                sh.Q[(2 * pa) * EIG_VP + jp] = cb * q0p - sb * q0q;
                sh.Q[(2 * pa) * EIG_VP + jq] = sb * q0p + cb * q0q;
                sh.Q[(2 * pa + 1) * EIG_VP + jp] = cb * q1p - sb * q1q;
                sh.Q[(2 * pa + 1) * EIG_VP + jq] = sb * q1p + cb * q1q;
                wave_sync_lds();
            }
        }
        if (lane < EIG_B) sh.theta[lane] = sh.H[lane * EIG_VP + lane];
#ifdef EIG_STAMPS
        if (lane == 0 && blockIdx.x == 0) g_eig_stamps[20] = sweep_count;
#endif
    }
    __syncthreads();
}

// One Rayleigh-Ritz + orthonormalisation step on the R x 16 block X = G V (LDS), in place:
//     S = X^T X = V^T G^2 V            (MFMA)
//     S = P D P^T                      (one 16 x 16 Jacobi)
//     X <- X P D^-1/2                  (MFMA; columns = orthonormal Ritz vectors of G^2 on span(V), images under G)
//     polish: X <- X (1.5 I - 0.5 X^T X) until ||X^T X - I||_max <= 2e-15   (Newton-Schulz, MFMA)
// sqrt(D_i) are Ritz values of G (from G^2 on the same subspace: lower bounds, second-order accurate like
// the ones of V^T G V); sh.top4 = sum of the four largest.  Directions with D_i <= 1e-28 D_max are dead
// (zero columns: exactly singular input, or R < 16).  No Cholesky, no serial 16-step chains, and no
// data-dependent fallback path.  When `values_only` the block is left untouched after the Jacobi.
__device__ __forceinline__ void ritz_orth16(double* X, int Rp, EigShared& sh) {
    gram16(X, Rp, sh, sh.H);
    jacobi16(sh);
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
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) pick += __shfl_xor(pick, d, 64);
            if (lane == 0) sh.top4 = pick;
        }
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            const double dj = sh.theta[j];
            const double rj = (dj > 1e-28 * dmax && dj > 0) ? rsqrt_nr(dj) : 0.0;
            sh.T[i * EIG_VP + j] = sh.Q[i * EIG_VP + j] * rj;
        }
        __syncthreads();
    }
    rowmul16(X, Rp, sh.T);
    for (int iter = 0; iter < 10; ++iter) {
        gram16(X, Rp, sh, sh.H);
        double err = 0;
        if (threadIdx.x < 256) {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            const double sv = sh.H[i * EIG_VP + j];
            const double target = (i == j && sh.H[i * EIG_VP + i] > 0.25) ? 1.0 : 0.0;  // dead columns stay 0
            err = fabs(sv - target);
            sh.T[i * EIG_VP + j] = (i == j ? 1.5 : 0.0) - 0.5 * sv;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) err = fmax(err, __shfl_xor(err, d, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = err;
        __syncthreads();
        double emx = 0;
        for (int k = 0; k < EIG_WAVES; ++k) emx = fmax(emx, sh.red[k]);
        __syncthreads();
#ifdef EIG_STAMPS
        if (threadIdx.x == 0 && blockIdx.x == 0) g_eig_stamps[21] = iter;
#endif
        if (emx <= 2e-15) break;
        rowmul16(X, Rp, sh.T);
    }
}

struct EigState {
    double trace, prev_sum, prev_delta, top4;
    int it, done, R, pad;
};

#define EIG_NFAST 5

__device__ __forceinline__ double block_sum(double v, EigShared& sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
__device__ __forceinline__ bool update_convergence(double s4, int it, double& prev_sum, double& prev_delta) {
    bool conv = false;
    const double delta = fabs(s4 - prev_sum);
    if (it >= 2) {
        double ratio = prev_delta > 0 ? delta / prev_delta : 0.0;
        ratio = fmin(fmax(ratio, 0.0), 0.9999);
        const double tail = delta * ratio / (1.0 - ratio);
        if (delta <= 4e-16 * s4 || (it >= 3 && tail <= 1e-14 * s4)) conv = true;
    }
    prev_delta = delta;
    prev_sum = s4;
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

// ---- per split: trace, degenerate cases, start block -------------------------------------------------
__global__ __launch_bounds__(EIG_THREADS) void k_eig_init(const SplitDev* __restrict__ splits,
                                                          const int2* __restrict__ dims,
                                                          const double* __restrict__ grams,
                                                          EigState* __restrict__ states, double* __restrict__ vt_pool,
                                                          double* __restrict__ scores, int* __restrict__ status,
                                                          const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + ((sizeof(EigShared) + 15) & ~(size_t)15));
    const int sid = order[blockIdx.x];  // heaviest splits first
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const int Rp = (R + 15) & ~15;
    const double* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;

    // trace(G) = sum of all squared singular values (exact for integer counts)
    double tr = 0;
    for (int i = threadIdx.x; i < R; i += EIG_THREADS) tr += G[(int64_t)i * gp + i];
    tr = block_sum(tr, sh);
    EigState st;
    st.trace = tr; st.prev_sum = 0; st.prev_delta = 0; st.top4 = 0; st.it = 0; st.done = 0; st.R = R; st.pad = 0;
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
    // Start block: fixed pseudo-random entries (robust for any matrix) plus a strong component on the
    // rows with the largest diagonal of G - for count matrices the dominant singular vectors sit on
    // the rows of the few very frequent patterns - then Cholesky-QR x2.
    double* dg = sh.part;  // R <= 1024 doubles fit in the 8 KiB reduction buffer
    for (int i = threadIdx.x; i < Rp; i += EIG_THREADS) dg[i] = i < R ? G[(int64_t)i * gp + i] : -1.0;
    __syncthreads();
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) {
        const int row = e >> 4, col = e & 15;
        V[row * EIG_VP + col] = row < R ? 0.02 * hash_unit(row, col) : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += EIG_THREADS) {
        const double di = dg[i];
        int rank = 0;
        for (int j = 0; j < R && rank < EIG_B; ++j) {
            const double dj = dg[j];
            rank += (dj > di || (dj == di && j < i)) ? 1 : 0;
        }
        if (rank < EIG_B) V[i * EIG_VP + rank] += 1.0;
    }
    __syncthreads();
    ritz_orth16(V, Rp, sh);  // here only as an orthonormaliser (the start block is not G times anything)
    store_vt(V, Rp, vt_pool + sp.ev_off, sp.rcap);
    if (threadIdx.x == 0) states[sid] = st;
}

// ---- Y = G V for every split: one workgroup per (split, 64-row block) ---------------------------------
// Per wave one 16-row tile; per 16-k block a lane loads 4 consecutive doubles of its G row (32 B: a wave
// reads 16 rows x 128 B) and 4 consecutive doubles of its V column, and issues 4 MFMAs; MFMA j
// sums k in {k0 + 4g + j : g = 0..3}, so the 16 k's are covered exactly once.
__global__ __launch_bounds__(256) void k_eig_gv(const GramItem* __restrict__ items,
                                                const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                const double* __restrict__ grams,
                                                const EigState* __restrict__ states,
                                                const double* __restrict__ vt_pool, double* __restrict__ y_pool) {
    const GramItem it = items[blockIdx.x];
    const int sid = it.sid;
    if (states[sid].done) return;
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const int Rp = (R + 15) & ~15;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int row0 = it.ti * 64 + w * 16;
    if (row0 >= Rp) return;
    const double* __restrict__ g = grams + sp.g_off + (int64_t)(row0 + fr) * sp.g_pitch + 4 * fk;
    const double* __restrict__ v = vt_pool + sp.ev_off + (int64_t)fr * sp.rcap + 4 * fk;
    double4_t acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int k0 = 0; k0 < Rp; k0 += 16) {
        const double2 a01 = *reinterpret_cast<const double2*>(g + k0);
        const double2 a23 = *reinterpret_cast<const double2*>(g + k0 + 2);
        const double2 b01 = *reinterpret_cast<const double2*>(v + k0);
        const double2 b23 = *reinterpret_cast<const double2*>(v + k0 + 2);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b01.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b01.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b23.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b23.y, acc, 0, 0, 0);
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
    double* V = reinterpret_cast<double*>(smem_raw + ((sizeof(EigShared) + 15) & ~(size_t)15));
    const int sid = order[blockIdx.x];
    EigState st = states[sid];
    if (st.done) return;
    const SplitDev& sp = splits[sid];
    const int R = st.R;
    const int Rp = (R + 15) & ~15;
    double* __restrict__ Vt = vt_pool + sp.ev_off;
    const double* __restrict__ Y = y_pool + sp.ev_off;
    const int vp = sp.rcap;
    STAMP(0);
    // Y -> LDS working block
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) V[(e >> 4) * EIG_VP + (e & 15)] = Y[e];
    __syncthreads();
    STAMP(1);
    ritz_orth16(V, Rp, sh);  // Ritz values (sh.top4) + next orthonormal block in one pass
    STAMP(2);
    st.it += 1;
    st.top4 = sh.top4;
    const bool conv = update_convergence(sh.top4, st.it, st.prev_sum, st.prev_delta);
    if (conv) {
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
__global__ __launch_bounds__(EIG_THREADS) void k_eig_finish(const SplitDev* __restrict__ splits,
                                                            const double* __restrict__ grams,
                                                            EigState* __restrict__ states,
                                                            const double* __restrict__ vt_pool,
                                                            double* __restrict__ scores, int* __restrict__ status,
                                                            const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + ((sizeof(EigShared) + 15) & ~(size_t)15));
    const int sid = order[blockIdx.x];
    EigState st = states[sid];
    if (st.done) return;
    const SplitDev& sp = splits[sid];
    const int R = st.R;
    const int Rp = (R + 15) & ~15;
    const double* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    load_vt(V, Rp, vt_pool + sp.ev_off, sp.rcap);
    __syncthreads();
    const int ntile = Rp >> 4;
    int converged = 0;
    while (st.it < EIG_MAXIT) {
        // ---- Y = G V : wave w owns row tiles w, w+8, ... ; k outer so one V fragment feeds all tiles
        double4_t acc[EIG_MAXT];
#pragma unroll
        for (int t = 0; t < EIG_MAXT; ++t) acc[t] = (double4_t){0, 0, 0, 0};
        for (int k0 = 0; k0 < Rp; k0 += 4) {
            const double b = V[(k0 + fk) * EIG_VP + fr];
            const double* __restrict__ gk = G + (int64_t)(k0 + fk) * gp + fr;  // G[k][row] == G[row][k]
#pragma unroll
            for (int t = 0; t < EIG_MAXT; ++t) {
                const int tile = w + t * EIG_WAVES;
                if (tile < ntile) {
                    const double a = gk[tile * 16];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                }
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
        __syncthreads();
        ritz_orth16(V, Rp, sh);
        st.it += 1;
        st.top4 = sh.top4;
        if (update_convergence(sh.top4, st.it, st.prev_sum, st.prev_delta)) {
            converged = 1;
            break;
        }
    }
    if (threadIdx.x == 0) {
        st.done = 1;
        states[sid] = st;
        write_score(st.top4, st.trace, st.it, converged != 0, scores, status, sid);
    }
}

int launch_eigen(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                 const double* grams, const GramItem* rowblocks_dev, int64_t n_rowblocks, const int* order_dev,
                 double* scores, int* status) {
    if (splits.empty()) return SP_OK;
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
    const size_t lds = ((sizeof(EigShared) + 15) & ~(size_t)15) + (size_t)maxr * EIG_VP * sizeof(double);
    static size_t attr = 0;
    if (lds > attr) {
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_init),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_rr), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_finish),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(k_eig_init, dim3((unsigned)S), dim3(EIG_THREADS), lds, ctx->stream, splits_dev, dims, grams,
                       states, vt, scores, status, order_dev);
    for (int round = 0; round < EIG_NFAST; ++round) {
        hipLaunchKernelGGL(k_eig_gv, dim3((unsigned)n_rowblocks), dim3(256), 0, ctx->stream, rowblocks_dev, splits_dev,
                           dims, grams, states, vt, yp);
        hipLaunchKernelGGL(k_eig_rr, dim3((unsigned)S), dim3(EIG_THREADS), lds, ctx->stream, splits_dev, dims, states,
                           vt, yp, scores, status, order_dev);
    }
    hipLaunchKernelGGL(k_eig_finish, dim3((unsigned)S), dim3(EIG_THREADS), lds, ctx->stream, splits_dev, grams, states,
                       vt, scores, status, order_dev);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

#ifdef EIG_STAMPS
extern "C" int sp_debug_eig_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eig_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : 2;
}
#endif
