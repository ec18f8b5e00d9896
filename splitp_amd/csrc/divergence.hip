// Mutual-information score of a flattening: the Kullback-Leibler divergence between the flattening and its rank-1
// approximation (outer product of its marginals).
//
// Replaces splitp/phylogenetics.py:332-341 (flattening_rank_1_approximation: r = column sums, c = row sums) and
// :364-373 (flattening_rank_1_approximation_divergence: sum over the non-zero cells of f * log(f / (r[y] * c[x])))),
// the scorer erickson_SVD uses with Method.mutual_information (phylogenetics.py:135-140).
//
// Batched form (pattern table + split list): a cell of the flattening is one pattern, so no matrix is built.  The
// reindex kernel of the dense route gives every pattern its compact (row, col) per split; the marginals are sums of
// counts grouped by row / by column - integer atomics for count tables (exact, order independent), fp64 atomics for
// float-weight tables - and the score is a sum over the D patterns, reduced in a fixed order.
// HBM-bound integer / byte work: 8 bytes of coordinates + 4 of count per pattern and split, three passes.
#include "common.h"

template <bool EXACT>
__global__ __launch_bounds__(256) void k_div_accum(int64_t D, const u32* __restrict__ rr, const u32* __restrict__ cc,
                                                    const u32* __restrict__ counts, const double* __restrict__ weights,
                                                    unsigned long long* __restrict__ marg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t s = blockIdx.y;
    if (i >= D) return;
    const u32 r = rr[s * D + i], c = cc[s * D + i];
    unsigned long long* rs = marg + s * 2 * D;
    unsigned long long* cs = rs + D;
    if (EXACT) {
        const unsigned long long v = counts[i];
        atomicAdd(rs + r, v);
        atomicAdd(cs + c, v);
    } else {
        const double w = weights[i];
        atomicAdd(reinterpret_cast<double*>(rs) + r, w);
        atomicAdd(reinterpret_cast<double*>(cs) + c, w);
    }
}

// one workgroup per split: sum_i f_i log(f_i / (rowsum * colsum)), fixed reduction tree
template <bool EXACT>
__global__ __launch_bounds__(256) void k_div_sum(int64_t D, const u32* __restrict__ rr, const u32* __restrict__ cc,
                                                  const u32* __restrict__ counts, const double* __restrict__ weights,
                                                  const unsigned long long* __restrict__ marg, double n_total,
                                                  double* __restrict__ out) {
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    const unsigned long long* rs = marg + s * 2 * D;
    const unsigned long long* cs = rs + D;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < D; i += 256) {
        const u32 r = rr[s * D + i], c = cc[s * D + i];
        if (EXACT) {
            const double v = (double)counts[i];
            if (v != 0.0) acc += (v / n_total) * log(v * n_total / ((double)rs[r] * (double)cs[c]));
        } else {
            const double w = weights[i];
            if (w != 0.0)
                acc += w * log(w / (reinterpret_cast<const double*>(rs)[r] * reinterpret_cast<const double*>(cs)[c]));
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[s] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Count tables up to 16 k patterns with fewer than 2^32 sites: marginals and sum in ONE kernel, one workgroup per split,
// the two marginal arrays (u32, indexed by compact row / column) in LDS.  The global-memory form above spends its time on
// 2 D S device atomics and three passes (0.46 ms for 501 splits of the 8.2 k-pattern table); here the atomics are LDS
// atomics of one workgroup and the coordinates are read from L2 twice.  Same terms, same 256-thread summation tree as
// k_div_sum: identical results.
__global__ __launch_bounds__(256) void k_div_fused(int64_t D, const u32* __restrict__ rr, const u32* __restrict__ cc,
                                                   const u32* __restrict__ counts, double n_total,
                                                   double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u32 marg_lds[];
    __shared__ double red[4];
    const int64_t s = blockIdx.x;
    u32* rs = marg_lds;
    u32* cs = marg_lds + D;
    const u32* __restrict__ rrow = rr + s * D;
    const u32* __restrict__ crow = cc + s * D;
    for (int64_t i = threadIdx.x; i < 2 * D; i += 256) marg_lds[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < D; i += 256) {
        const u32 v = counts[i];
        atomicAdd(rs + rrow[i], v);
        atomicAdd(cs + crow[i], v);
    }
    __syncthreads();
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < D; i += 256) {
        const double v = (double)counts[i];
        if (v != 0.0) acc += (v / n_total) * log(v * n_total / ((double)rs[rrow[i]] * (double)cs[crow[i]]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[s] = (red[0] + red[1]) + (red[2] + red[3]);
}

int launch_divergence(sp_ctx* ctx, bool exact, int64_t D, int64_t S, const u32* rr, const u32* cc, const u32* counts,
                      const double* weights, double n_total, unsigned long long* marg, double* out) {
    if (S == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_DIVERGENCE);
    if (exact && D <= 16384 && n_total < 4294967296.0 && !ctx->opt.divergence_global) {
        const size_t lds = (size_t)2 * D * 4;
        static PerDeviceOnce attr;
        if (attr.need(ctx->device)) {
            SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_div_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       131072));
            attr.done(ctx->device);
        }
        hipLaunchKernelGGL(k_div_fused, dim3((unsigned)S), dim3(256), lds, ctx->stream, D, rr, cc, counts, n_total, out);
        SP_HIP(hipGetLastError());
        return SP_OK;
    }
    SP_HIP(hipMemsetAsync(marg, 0, (size_t)S * 2 * (size_t)D * 8, ctx->stream));
    const dim3 grid((unsigned)((D + 255) / 256), (unsigned)S);
    if (exact) {
        hipLaunchKernelGGL(k_div_accum<true>, grid, dim3(256), 0, ctx->stream, D, rr, cc, counts, weights, marg);
        hipLaunchKernelGGL(k_div_sum<true>, dim3((unsigned)S), dim3(256), 0, ctx->stream, D, rr, cc, counts, weights, marg,
                           n_total, out);
    } else {
        hipLaunchKernelGGL(k_div_accum<false>, grid, dim3(256), 0, ctx->stream, D, rr, cc, counts, weights, marg);
        hipLaunchKernelGGL(k_div_sum<false>, dim3((unsigned)S), dim3(256), 0, ctx->stream, D, rr, cc, counts, weights, marg,
                           n_total, out);
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// ---- generic matrix form (flattening_rank_1_approximation_divergence(matrix)) -----------------------------------
// row sums: one thread per row; column sums: one thread per column (coalesced); both in index order like the
// reference's Python sums, so the marginals are the reference's bit for bit.
__global__ void k_divm_rows(const double* __restrict__ m, int64_t rows, int64_t cols, double* __restrict__ rowsum) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int64_t c = 0; c < cols; ++c) s += m[r * cols + c];
    rowsum[r] = s;
}
__global__ void k_divm_cols(const double* __restrict__ m, int64_t rows, int64_t cols, double* __restrict__ colsum) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int64_t r = 0; r < rows; ++r) s += m[r * cols + c];
    colsum[c] = s;
}
// partial[r] = sum over the row (column order); total = sum over the rows in order (one thread: the reference's order)
__global__ void k_divm_terms(const double* __restrict__ m, int64_t rows, int64_t cols, const double* __restrict__ rowsum,
                             const double* __restrict__ colsum, double* __restrict__ partial) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int64_t c = 0; c < cols; ++c) {
        const double f = m[r * cols + c];
        if (f != 0.0) s += f * log(f / (colsum[c] * rowsum[r]));
    }
    partial[r] = s;
}
__global__ void k_divm_total(const double* __restrict__ partial, int64_t rows, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int64_t r = 0; r < rows; ++r) s += partial[r];
        *out = s;
    }
}

int launch_divergence_matrix(sp_ctx* ctx, const double* m_dev, int64_t rows, int64_t cols, double* scratch, double* out) {
    PhaseScope ps(ctx, SP_PHASE_DIVERGENCE);
    double* rowsum = scratch;
    double* colsum = scratch + rows;
    double* partial = colsum + cols;
    hipLaunchKernelGGL(k_divm_rows, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, ctx->stream, m_dev, rows, cols, rowsum);
    hipLaunchKernelGGL(k_divm_cols, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, ctx->stream, m_dev, rows, cols, colsum);
    hipLaunchKernelGGL(k_divm_terms, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, ctx->stream, m_dev, rows, cols, rowsum,
                       colsum, partial);
    hipLaunchKernelGGL(k_divm_total, dim3(1), dim3(64), 0, ctx->stream, partial, rows, out);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
