// Probe for round 5 (GPU box): what ONE WAVE needs for an iteration of the sparse kernel's Gram path - Y = G V on a dense
// R x R Gram matrix in LDS (R = 16: 2-taxon row side, R = 64: 3-taxon row side), Ritz sum trace(V^T Y), 4 x 4 Gram matrix
// Y^T Y, Cholesky factor, V = Y L^-T - with no workgroup barrier anywhere: the other 15 waves of the 1024-thread workgroup
// wait at ONE barrier behind the loop.  k_sparse_score runs this iteration block-wide today: 6.3 k (R = 16) and 8.1 k
// (R = 64) cycles per iteration at config 2 (profiles/r03_stamps_sparse_kernels.txt: product 1.6 - 3.3 k, sum + Gram 2.1 k,
// Cholesky + stop rule 1.9 k, orth 0.7 k), four iterations a split, 165 of 501 splits.
// Measured (profiles/r04_gram_iter_probe.txt): R = 16: 2.1 k ticks per iteration (a third of today's 6.3 k); R = 64: 7.9 k -
// no gain: 64 k-steps of (one b64 + two broadcast b128 LDS reads + 4 FMA) cost ~94 ticks each in a single wave, LDS latency
// with nothing to hide it - and still 67 with the loads of 8 or 16 steps in flight (6.2 k per iteration: a broadcast b128
// read occupies the LDS as long as a scattered one); there the block-wide product (3.3 k) has to stay and only the 4.7 k of sum + Gram + Cholesky +
// orth behind it can move into one wave (~1.8 k by the R = 16 figure).  Worth 45 x 17 k + 120 x 8 k = 1.7 M of the launch's
// 61 M CU-cycles (2.8 %: 0.0954 -> ~0.093 ms) - not the 0.090 target on its own.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/gram_iter_probe.hip -o /tmp/gram_iter_probe && /tmp/gram_iter_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define DPP_QUAD_XOR1 0xB1
#define DPP_QUAD_XOR2 0x4E
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_SHR8 0x118

template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// sum over the lanes that hold DISTINCT rows, the same value in every lane afterwards; fixed order.
template <int R>
__device__ __forceinline__ double all_sum(double x) {
    x += dpp<DPP_QUAD_XOR1>(x);
    x += dpp<DPP_QUAD_XOR2>(x);
    x += dpp<DPP_ROW_SHR4>(x);
    x += dpp<DPP_ROW_SHR8>(x);                 // lane 15 of every 16-lane row: the row's sum
    if (R == 64) {
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        return lane_value(x, 63);
    }
    return lane_value(x, 15);                   // R = 16: the rows live in lanes 0..15 (lanes 16..63 hold replicas)
}

template <int R, int BATCH>
__global__ __launch_bounds__(1024) void k_probe(const double* __restrict__ Gin, double* __restrict__ sums, long long* __restrict__ cyc,
                                                int iters) {
    __shared__ double G[R * R];        // symmetric: G[k * R + i] = G[i][k], consecutive lanes read consecutive words
    __shared__ double V[R * 4];        // [row][4]
    for (int i = threadIdx.x; i < R * R; i += blockDim.x) G[i] = Gin[i];
    for (int i = threadIdx.x; i < R * 4; i += blockDim.x) {
        const int row = i >> 2, c = i & 3;
        V[i] = (row % 4 == c ? 1.0 : 0.0) + 0.01 * (double)((row * 7 + c * 3) % 11);     // full rank, not orthonormal
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int row = R == 64 ? lane : (lane & 15);
        const int q = R == 64 ? 0 : (lane >> 4);                     // R = 16: four k-quarters per row
        constexpr int KQ = R == 64 ? 64 : 4;
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            double y[4] = {0, 0, 0, 0};
            if (BATCH > 1 && R == 64) {
                // all loads of BATCH k-steps in flight before the first FMA (the plain loop above leaves two k-steps in
                // flight and pays an LDS round trip for every pair: 94 ticks a step)
#pragma unroll 1
                for (int k0 = 0; k0 < 64; k0 += BATCH) {
                    double g[BATCH];
                    double2 lo[BATCH], hi[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        g[u] = G[(k0 + u) * R + row];
                        lo[u] = *reinterpret_cast<const double2*>(&V[(k0 + u) * 4]);
                        hi[u] = *reinterpret_cast<const double2*>(&V[(k0 + u) * 4 + 2]);
                    }
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        y[0] = fma(g[u], lo[u].x, y[0]);
                        y[1] = fma(g[u], lo[u].y, y[1]);
                        y[2] = fma(g[u], hi[u].x, y[2]);
                        y[3] = fma(g[u], hi[u].y, y[3]);
                    }
                }
            } else {
#pragma unroll 8
                for (int kk = 0; kk < KQ; ++kk) {
                    const int k = q * KQ + kk;
                    const double g = G[k * R + row];
                    const double2 lo = *reinterpret_cast<const double2*>(&V[k * 4]), hi = *reinterpret_cast<const double2*>(&V[k * 4 + 2]);
                    y[0] = fma(g, lo.x, y[0]);
                    y[1] = fma(g, lo.y, y[1]);
                    y[2] = fma(g, hi.x, y[2]);
                    y[3] = fma(g, hi.y, y[3]);
                }
            }
            if (R == 16) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    y[c] += __shfl_xor(y[c], 16, 64);
                    y[c] += __shfl_xor(y[c], 32, 64);
                }
            }
            const double2 vlo = *reinterpret_cast<const double2*>(&V[row * 4]), vhi = *reinterpret_cast<const double2*>(&V[row * 4 + 2]);
            const double ritz = all_sum<R>(fma(vlo.x, y[0], fma(vlo.y, y[1], fma(vhi.x, y[2], vhi.y * y[3]))));
            double S[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b) S[a][b] = all_sum<R>(y[a] * y[b]);
            // Cholesky S = L L^T in every lane (uniform values), reciprocal pivots
            double L[4][4], inv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double d = S[j][j];
#pragma unroll
                for (int p = 0; p < j; ++p) d = fma(-L[j][p], L[j][p], d);
                const double r = rsqrt(d);
                inv[j] = r;
                L[j][j] = d * r;
#pragma unroll
                for (int i = j + 1; i < 4; ++i) {
                    double s = S[j][i];
#pragma unroll
                    for (int p = 0; p < j; ++p) s = fma(-L[i][p], L[j][p], s);
                    L[i][j] = s * r;
                }
            }
            // V = Y L^-T (forward substitution per row)
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double s = y[j];
#pragma unroll
                for (int p = 0; p < j; ++p) s = fma(-L[j][p], v[p], s);
                v[j] = s * inv[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // every lane has read its V rows
            if (q == 0) {
                *reinterpret_cast<double2*>(&V[row * 4]) = double2{v[0], v[1]};
                *reinterpret_cast<double2*>(&V[row * 4 + 2]) = double2{v[2], v[3]};
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // ... and sees the new block
            if (lane == 0) sums[it] = ritz;
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[0] = t1 - t0;
    }
    __syncthreads();        // the one barrier the other 15 waves wait at
}

template <int R, int BATCH>
static void run(const char* what) {
    // G = C C^T with a decaying spectrum (a flattening's row Gram matrix looks like this: a few large, many small)
    const int K = 300;
    std::vector<double> Cm(R * K), G(R * R, 0.0);
    unsigned x = 4242u + R;
    for (int i = 0; i < R; ++i)
        for (int k = 0; k < K; ++k) {
            x = x * 1664525u + 1013904223u;
            const double u = (double)(x >> 8) / (double)(1 << 24) - 0.5;
            Cm[i * K + k] = u * std::pow(0.93, k) * (k < 4 ? 6.0 : 1.0);
        }
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += Cm[i * K + k] * Cm[j * K + k];
            G[i * R + j] = s;
        }
    const int iters_max = 8;
    // host reference of the same iteration
    std::vector<double> V(R * 4), Y(R * 4), ref(iters_max);
    for (int i = 0; i < R * 4; ++i) {
        const int row = i >> 2, c = i & 3;
        V[i] = (row % 4 == c ? 1.0 : 0.0) + 0.01 * (double)((row * 7 + c * 3) % 11);
    }
    for (int it = 0; it < iters_max; ++it) {
        for (int i = 0; i < R; ++i)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int k = 0; k < R; ++k) s += G[i * R + k] * V[k * 4 + c];
                Y[i * 4 + c] = s;
            }
        double ritz = 0, S[4][4] = {{0}}, L[4][4] = {{0}};
        for (int i = 0; i < R; ++i)
            for (int c = 0; c < 4; ++c) ritz += V[i * 4 + c] * Y[i * 4 + c];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                for (int i = 0; i < R; ++i) S[a][b] += Y[i * 4 + a] * Y[i * 4 + b];
        for (int j = 0; j < 4; ++j) {
            double d = S[j][j];
            for (int p = 0; p < j; ++p) d -= L[j][p] * L[j][p];
            L[j][j] = std::sqrt(d);
            for (int i = j + 1; i < 4; ++i) {
                double s = S[i][j];
                for (int p = 0; p < j; ++p) s -= L[i][p] * L[j][p];
                L[i][j] = s / L[j][j];
            }
        }
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = Y[i * 4 + j];
                for (int p = 0; p < j; ++p) s -= L[j][p] * V[i * 4 + p];
                V[i * 4 + j] = s / L[j][j];
            }
        ref[it] = ritz;
    }
    double *dG, *dS;
    long long* dC;
    hipMalloc(&dG, R * R * 8);
    hipMalloc(&dS, iters_max * 8);
    hipMalloc(&dC, 8);
    hipMemcpy(dG, G.data(), R * R * 8, hipMemcpyHostToDevice);
    long long c4 = 0, c8 = 0;
    std::vector<double> got(iters_max);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_probe<R, BATCH>), dim3(1), dim3(1024), 0, 0, dG, dS, dC, 4);
        hipMemcpy(&c4, dC, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL((k_probe<R, BATCH>), dim3(1), dim3(1024), 0, 0, dG, dS, dC, 8);
        hipMemcpy(&c8, dC, 8, hipMemcpyDeviceToHost);
    }
    hipMemcpy(got.data(), dS, iters_max * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int it = 1; it < iters_max; ++it) worst = std::fmax(worst, std::fabs(got[it] - ref[it]) / std::fabs(ref[it]));
    printf("%s: 4 iterations %lld ticks, 8 iterations %lld ticks -> %.0f ticks per iteration (one wave, no barrier); "
           "Ritz sums against the host's: max relative difference %.2e (last sum %.12g)\n",
           what, c4, c8, (double)(c8 - c4) / 4.0, worst, got[iters_max - 1]);
    hipFree(dG);
    hipFree(dS);
    hipFree(dC);
}

int main() {
    run<16, 1>("R = 16 (2-taxon row side; block-wide today: ~6.3 k cycles per iteration)");
    run<64, 1>("R = 64 (3-taxon row side; block-wide today: ~8.1 k cycles per iteration), plain loop");
    run<64, 8>("R = 64, loads of 8 k-steps in flight");
    run<64, 16>("R = 64, loads of 16 k-steps in flight");
    return 0;
}
