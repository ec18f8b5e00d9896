// Top-4 eigenvalues of the Gram matrix + the split score, one workgroup per split.
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
// 1e-14 relative).  Phylogenetic flattenings have lambda_17 / lambda_4 < 1e-3, so this takes 3-5
// products; arbitrary matrices are handled by the same loop, capped at EIG_MAXIT (status bit 0).
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

// Cholesky of the 16 x 16 SPD matrix in sh.H (lower triangle into sh.L) by the first 16 lanes of
// wave 0.  A column whose pivot collapses (numerically dependent or zero direction) is marked dead:
// it is zeroed by the solve below.  Ends with a barrier.
__device__ __forceinline__ void chol16(EigShared& sh) {
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double dmax = 0;
        for (int j = 0; j < EIG_B; ++j) dmax = fmax(dmax, sh.H[j * EIG_VP + j]);
        for (int j = 0; j < EIG_B; ++j) {
            // pivot
            double d = sh.H[j * EIG_VP + j];
            const double d0 = d;
            for (int k = 0; k < j; ++k) {
                const double l = sh.L[j * EIG_VP + k];
                d -= l * l;
            }
            const bool dead = !(d > 1e-13 * d0) || !(d0 > 1e-28 * dmax);
            const double piv = dead ? 1.0 : sqrt(d);
            if (i == 0) {
                sh.dead[j] = dead;
                sh.L[j * EIG_VP + j] = piv;
            }
            if (i > j && i < EIG_B) {
                double v = sh.H[i * EIG_VP + j];
                for (int k = 0; k < j; ++k) v -= sh.L[i * EIG_VP + k] * sh.L[j * EIG_VP + k];
                sh.L[i * EIG_VP + j] = dead ? 0.0 : v / piv;
            }
            wave_sync_lds();
        }
        // T = L^-T: lane c (< 16) solves L w = e_c by forward substitution (w kept in LDS, rolled
        // loops: unrolling makes the compiler hoist all of L into registers), T[c][r] = W[r][c]
        if (i < EIG_B) {
            double* wv = sh.part + i * EIG_B;
#pragma unroll 1
            for (int r = 0; r < EIG_B; ++r) {
                double v = (r == i) ? 1.0 : 0.0;
#pragma unroll 1
                for (int k = 0; k < r; ++k) v -= sh.L[r * EIG_VP + k] * wv[k];
                wv[r] = v / sh.L[r * EIG_VP + r];
            }
#pragma unroll 1
            for (int r = 0; r < EIG_B; ++r) sh.T[i * EIG_VP + r] = (sh.dead[r] || sh.dead[i]) ? 0.0 : wv[r];
        }
    }
    __syncthreads();
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

// Eigen-decomposition of the symmetric 16 x 16 matrix sh.H by parallel-order (round-robin) Jacobi,
// wave 0 only: H -> diagonal (sh.theta), eigenvectors in the columns of sh.Q.  Ends with a barrier.
__device__ __forceinline__ void jacobi16(EigShared& sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        for (int e = lane; e < EIG_B * EIG_B; e += 64) sh.Q[(e >> 4) * EIG_VP + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        wave_sync_lds();
        for (int sweep = 0; sweep < 12; ++sweep) {
            // convergence: off-diagonal mass against the diagonal
            double off = 0, dg = 0;
            for (int e = lane; e < EIG_B * EIG_B; e += 64) {
                const double v = sh.H[(e >> 4) * EIG_VP + (e & 15)];
                if ((e >> 4) == (e & 15)) dg += v * v; else off += v * v;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                off += __shfl_xor(off, d, 64);
                dg += __shfl_xor(dg, d, 64);
            }
            if (!(off > 1e-30 * dg)) break;
            for (int round = 0; round < EIG_B - 1; ++round) {
                // 8 disjoint pairs; lanes 8*i .. 8*i+7 serve pair i
                const int pi = lane >> 3, sub = lane & 7;
                int p, q;
                if (pi == 0) {
                    p = EIG_B - 1;
                    q = round;
                } else {
                    p = (round + pi) % (EIG_B - 1);
                    q = (round + (EIG_B - 1) - pi) % (EIG_B - 1);
                }
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = sh.H[p * EIG_VP + p], aqq = sh.H[q * EIG_VP + q], apq = sh.H[p * EIG_VP + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-20 * sqrt(fabs(app * aqq)) ) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                }
                wave_sync_lds();
                // columns p,q of H and Q: rows sub, sub+8
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = sub + 8 * h;
                    const double hp = sh.H[i * EIG_VP + p], hq = sh.H[i * EIG_VP + q];
                    sh.H[i * EIG_VP + p] = c * hp - s * hq;
                    sh.H[i * EIG_VP + q] = s * hp + c * hq;
                    const double qp = sh.Q[i * EIG_VP + p], qq = sh.Q[i * EIG_VP + q];
                    sh.Q[i * EIG_VP + p] = c * qp - s * qq;
                    sh.Q[i * EIG_VP + q] = s * qp + c * qq;
                }
                wave_sync_lds();
                // rows p,q of H: cols sub, sub+8
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = sub + 8 * h;
                    const double hp = sh.H[p * EIG_VP + j], hq = sh.H[q * EIG_VP + j];
                    sh.H[p * EIG_VP + j] = c * hp - s * hq;
                    sh.H[q * EIG_VP + j] = s * hp + c * hq;
                }
                wave_sync_lds();
            }
        }
        // sum of the four largest eigenvalues: rank every diagonal entry inside the wave
        const double th = lane < EIG_B ? sh.H[lane * EIG_VP + lane] : -1e300;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < EIG_B; ++j) {
            const double o = __shfl(th, j, 64);
            rank += (o > th || (o == th && j < lane)) ? 1 : 0;
        }
        double pick = (lane < EIG_B && rank < 4) ? th : 0.0;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) pick += __shfl_xor(pick, d, 64);
        if (lane < EIG_B) sh.theta[lane] = th;
        if (lane == 0) sh.top4 = pick;
    }
    __syncthreads();
}

__global__ __launch_bounds__(EIG_THREADS) void k_eigen(const SplitDev* __restrict__ splits,
                                                       const int2* __restrict__ dims,
                                                       const double* __restrict__ grams, double* __restrict__ scores,
                                                       int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EigShared& sh = *reinterpret_cast<EigShared*>(smem_raw);
    double* V = reinterpret_cast<double*>(smem_raw + ((sizeof(EigShared) + 15) & ~(size_t)15));

    const int sid = blockIdx.x;
    const SplitDev& sp = splits[sid];
    const int R = min(dims[sid].x, sp.rcap);
    const int Rp = (R + 15) & ~15;
    const double* __restrict__ G = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;

    // trace(G) = sum of all squared singular values (exact for integer counts)
    double tr = 0;
    for (int i = threadIdx.x; i < R; i += EIG_THREADS) tr += G[(int64_t)i * gp + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    if (lane == 0) sh.red[w] = tr;
    __syncthreads();
    tr = 0;
    for (int i = 0; i < EIG_WAVES; ++i) tr += sh.red[i];
    __syncthreads();
    if (R <= 4 || !(tr > 0)) {
        // min(shape) <= 4: the reference computes 1 - x/x = 0 exactly; all-zero matrix: 0/0 = nan
        if (threadIdx.x == 0) {
            scores[sid] = (tr > 0) ? 0.0 : __builtin_nan("");
            status[sid] = 0;
        }
        return;
    }

    // start block: fixed pseudo-random entries (rows >= R zero), orthonormalised by Cholesky-QR x2
    for (int e = threadIdx.x; e < Rp * EIG_B; e += EIG_THREADS) {
        const int row = e >> 4, col = e & 15;
        V[row * EIG_VP + col] = row < R ? hash_unit(row, col) : 0.0;
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        gram16(V, Rp, sh, sh.H);
        chol16(sh);
        rowmul16(V, Rp, sh.T);
    }

    const int ntile = Rp >> 4;
    double prev_sum = 0, prev_delta = 0, top4 = 0;
    int it = 0, converged = 0;
    for (it = 1; it <= EIG_MAXIT; ++it) {
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
        // ---- H = V^T Y from the accumulators: reg r of a tile holds rows 4r..4r+3 -> a B operand
        double4_t hacc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < EIG_MAXT; ++t) {
            const int tile = w + t * EIG_WAVES;
            if (tile < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double a = V[(tile * 16 + 4 * r + fk) * EIG_VP + fr];
                    hacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[t][r], hacc, 0, 0, 0);
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
        reduce16(hacc, sh, sh.H);
        {  // symmetrise (V^T G V is symmetric up to rounding; Jacobi assumes exact symmetry)
            double a = 0, b2 = 0;
            const int i = (threadIdx.x >> 4) & 15, j = threadIdx.x & 15;
            if (threadIdx.x < 256) {
                a = sh.H[i * EIG_VP + j];
                b2 = sh.H[j * EIG_VP + i];
            }
            __syncthreads();
            if (threadIdx.x < 256) sh.H[i * EIG_VP + j] = 0.5 * (a + b2);
            __syncthreads();
        }
        jacobi16(sh);
        // ---- convergence of the sum of the four largest Ritz values (uniform across the block)
        {
            const double s4 = sh.top4;
            top4 = s4;
            const double delta = fabs(s4 - prev_sum);
            if (it >= 2) {
                double ratio = prev_delta > 0 ? delta / prev_delta : 0.0;
                ratio = fmin(fmax(ratio, 0.0), 0.9999);
                const double tail = delta * ratio / (1.0 - ratio);
                if (delta <= 4e-16 * s4 || (it >= 3 && tail <= 1e-14 * s4)) converged = 1;
            }
            prev_delta = delta;
            prev_sum = s4;
        }
        if (converged) break;
        // ---- Z = Y Q  (in place, MFMA)
        rowmul16(V, Rp, sh.Q);
        // ---- V = orth(Z): Cholesky-QR twice
        for (int pass = 0; pass < 2; ++pass) {
            gram16(V, Rp, sh, sh.H);
            chol16(sh);
            rowmul16(V, Rp, sh.T);
        }
    }
    if (threadIdx.x == 0) {
        const double op = 1.0 - top4 / tr;
        scores[sid] = sqrt(op > 0 ? op : 0.0);
        status[sid] = (converged ? 0 : 1) | (it << 8);
    }
}

int eigen_work_doubles_per_split(int rcap) { (void)rcap; return 0; }

int launch_eigen(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                 const double* grams, double* work, double* scores, int* status) {
    (void)work;
    if (splits.empty()) return SP_OK;
    int maxr = 0;
    for (const auto& s : splits) maxr = std::max(maxr, (int)s.rcap);
    SP_REQUIRE(maxr <= EIG_MAXR, SP_ELIMIT,
               "eigen kernel: the smaller side of a flattening has %d (padded) rows; this build keeps the iteration "
               "block in LDS and supports at most %d (n_taxa <= 11 on the dense route)", maxr, EIG_MAXR);
    PhaseScope ps(ctx, SP_PHASE_EIGEN);
    const size_t lds = ((sizeof(EigShared) + 15) & ~(size_t)15) + (size_t)maxr * EIG_VP * sizeof(double);
    static size_t attr = 0;
    if (lds > attr) {
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eigen), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(k_eigen, dim3((unsigned)splits.size()), dim3(EIG_THREADS), lds, ctx->stream, splits_dev, dims,
                       grams, scores, status);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
