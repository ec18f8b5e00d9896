// Flattening kernels: per-split re-indexing of the pattern table.
//
// Replaces the per-pattern Python loops of the reference:
//   splitp/constructions.py:37-45   (reduced: row/col index per pattern, nested dict write)
//   splitp/constructions.py:88-101  (sparse/dok: row/col index per pattern, dok assignment)
//   splitp/constructions.py:166-171 (__index_of, base-4 Horner, first char most significant)
//   splitp/constructions.py:48-54   (sorted used rows / cols -> compact dense matrix)
//
// Data model: the alignment lives in HBM as D (key, count|weight) pairs.  A split is a
// bit-permutation of the 2n-bit key into (row, col); because the split covers every taxon the map
// pattern -> (row, col) is collision-free, so a flattening is a pure scatter (no accumulation).
#include "common.h"

#define RX_THREADS 256

// shifts[i] = bit position of taxon taxa[i]'s digit inside the key, staged in LDS by load_shifts()
__device__ __forceinline__ void split_rowcol(u64 key, const int* shifts, int nr, int nc, u64& r, u64& c) {
    u64 rr = 0, cc = 0;
    for (int i = 0; i < nr; ++i) rr = (rr << 2) | ((key >> shifts[i]) & 3ull);
    for (int i = 0; i < nc; ++i) cc = (cc << 2) | ((key >> shifts[nr + i]) & 3ull);
    r = rr;
    c = cc;
}

__device__ __forceinline__ void load_shifts(int* shifts, const SplitDev& sp, int n) {
    if (threadIdx.x < 32) {
        const int t = threadIdx.x < sp.nr + sp.nc ? sp.taxa[threadIdx.x] : 0;
        shifts[threadIdx.x] = 2 * (n - 1 - t);
    }
    __syncthreads();
}

// exclusive scan of one u32 per thread over a 256-thread block (4 waves of 64)
__device__ __forceinline__ u32 block_excl_scan256(u32 v, u32* sh /*[5]*/, u32& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    u32 base = 0;
    for (int i = 0; i < w; ++i) base += sh[i];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + x - v;
}

// One workgroup per split:
//   1. presence bitmaps of the row keys and column keys (LDS when they fit, else the HBM pool)
//   2. popcount prefix per 64-bit word  -> rank of a key = prefix[word] + popc(bits below)
//   3. compact coordinates (rr, cc) of every pattern; dims[s] = (#used rows, #used cols)
// Ranks in ascending key order are exactly the reference's sorted(row keys) / sorted(used_cols)
// (constructions.py:48-49,:52).
// ARR = true: (row, col) come from explicit index arrays (COO input) instead of a key bit-gather.
template <bool LDS, bool ARR>
__global__ __launch_bounds__(RX_THREADS) void k_reindex(const u64* __restrict__ keys, int64_t D, int n,
                                                        const SplitDev* __restrict__ splits, u64* __restrict__ bm_pool,
                                                        u32* __restrict__ pf_pool, int2* __restrict__ dims,
                                                        u32* __restrict__ rr_out, u32* __restrict__ cc_out,
                                                        int lds_words, const int64_t* __restrict__ rows_in,
                                                        const int64_t* __restrict__ cols_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 sh_scan[5];
    __shared__ int shifts[32];
    const SplitDev& sp = splits[blockIdx.x];
    if (!ARR) load_shifts(shifts, sp, n);
    const int nr = sp.nr, nc = sp.nc, rw = sp.rw, cw = sp.cw;
    const int W = rw + cw;
    u64* bm;
    u32* pf;
    if (LDS) {
        bm = reinterpret_cast<u64*>(smem);
        pf = reinterpret_cast<u32*>(smem + (size_t)lds_words * 8);
    } else {
        bm = bm_pool + sp.bm_off;
        pf = pf_pool + sp.pfx_off;
    }
    for (int i = threadIdx.x; i < W; i += RX_THREADS) bm[i] = 0;
    __syncthreads();
    // 1. presence bits (test before the atomic: after the first few hundred patterns most bits are set)
    for (int64_t i = threadIdx.x; i < D; i += RX_THREADS) {
        u64 r, c;
        if (ARR) {
            r = (u64)rows_in[i];
            c = (u64)cols_in[i];
        } else {
            split_rowcol(keys[i], shifts, nr, nc, r, c);
        }
        const u64 rb = 1ull << (r & 63), cb = 1ull << (c & 63);
        u64* rp = bm + (r >> 6);
        u64* cp = bm + rw + (c >> 6);
        if (LDS) {
            if (!(*(volatile u64*)rp & rb)) atomicOr(rp, rb);
            if (!(*(volatile u64*)cp & cb)) atomicOr(cp, cb);
        } else {
            if (!(__hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & rb)) atomicOr(rp, rb);
            if (!(__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & cb)) atomicOr(cp, cb);
        }
    }
    __threadfence_block();
    __syncthreads();
    // 2. rank prefixes, rows then cols
    u32 totals[2];
    for (int which = 0; which < 2; ++which) {
        const int base = which ? rw : 0, cnt = which ? cw : rw;
        const int per = (cnt + RX_THREADS - 1) / RX_THREADS;
        const int lo = min(cnt, (int)threadIdx.x * per), hi = min(cnt, lo + per);
        u32 s = 0;
        for (int i = lo; i < hi; ++i) {
            u64 wv = LDS ? bm[base + i] : __hip_atomic_load(bm + base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s += __popcll(wv);
        }
        u32 tot;
        u32 run = block_excl_scan256(s, sh_scan, tot);
        for (int i = lo; i < hi; ++i) {
            u64 wv = LDS ? bm[base + i] : __hip_atomic_load(bm + base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pf[base + i] = run;
            run += __popcll(wv);
        }
        totals[which] = tot;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) dims[blockIdx.x] = make_int2((int)totals[0], (int)totals[1]);
    // 3. compact coordinates
    u32* rro = rr_out + (int64_t)blockIdx.x * D;
    u32* cco = cc_out + (int64_t)blockIdx.x * D;
    for (int64_t i = threadIdx.x; i < D; i += RX_THREADS) {
        u64 r, c;
        if (ARR) {
            r = (u64)rows_in[i];
            c = (u64)cols_in[i];
        } else {
            split_rowcol(keys[i], shifts, nr, nc, r, c);
        }
        u64 rwv, cwv;
        if (LDS) {
            rwv = bm[r >> 6];
            cwv = bm[rw + (c >> 6)];
        } else {
            rwv = __hip_atomic_load(bm + (r >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cwv = __hip_atomic_load(bm + rw + (c >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rro[i] = pf[r >> 6] + __popcll(rwv & ((1ull << (r & 63)) - 1));
        cco[i] = pf[rw + (c >> 6)] + __popcll(cwv & ((1ull << (c & 63)) - 1));
    }
}

int launch_reindex(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* splits_dev,
                   const std::vector<SplitDev>& splits, u64* bitmaps, u32* prefixes, int2* dims, u32* rr, u32* cc) {
    if (splits.empty()) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_REINDEX);
    // LDS path when every split's bitmaps + prefixes fit in 96 KiB: words*8 + words*4
    int maxw = 0;
    for (const auto& s : splits) maxw = std::max(maxw, s.rw + s.cw);
    const size_t lds_bytes = (size_t)maxw * 12;
    const int S = (int)splits.size();
    if (lds_bytes <= 96 * 1024) {
        static PerDeviceOnce attr_set;
        if (attr_set.need(ctx->device)) {
            SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_reindex<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_set.done(ctx->device);
        }
        hipLaunchKernelGGL((k_reindex<true, false>), dim3(S), dim3(RX_THREADS), lds_bytes, ctx->stream, keys, D, n_taxa,
                           splits_dev, bitmaps, prefixes, dims, rr, cc, maxw, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL((k_reindex<false, false>), dim3(S), dim3(RX_THREADS), 0, ctx->stream, keys, D, n_taxa,
                           splits_dev, bitmaps, prefixes, dims, rr, cc, 0, nullptr, nullptr);
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// COO input: compact the explicit (row, col) indices of one matrix (bitmaps in the HBM pool).
int launch_reindex_coo(sp_ctx* ctx, const int64_t* rows_in, const int64_t* cols_in, int64_t nnz,
                       const SplitDev* split_dev, u64* bitmaps, u32* prefixes, int2* dims, u32* rr, u32* cc) {
    PhaseScope ps(ctx, SP_PHASE_REINDEX);
    hipLaunchKernelGGL((k_reindex<false, true>), dim3(1), dim3(RX_THREADS), 0, ctx->stream, nullptr, nnz, 0, split_dev,
                       bitmaps, prefixes, dims, rr, cc, 0, rows_in, cols_in);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// ---- raw (row, col) of every pattern: the sparse/dok format, constructions.py:88-93 -----------
__global__ void k_bit_indices(const u64* __restrict__ keys, int64_t D, int n, const SplitDev* __restrict__ split,
                              int64_t* __restrict__ rows, int64_t* __restrict__ cols) {
    __shared__ int shifts[32];
    const SplitDev& sp = split[0];
    load_shifts(shifts, sp, n);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    u64 r, c;
    split_rowcol(keys[i], shifts, sp.nr, sp.nc, r, c);
    rows[i] = (int64_t)r;
    cols[i] = (int64_t)c;
}

int launch_bit_indices(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* split_dev, int64_t* rows,
                       int64_t* cols) {
    if (D == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_REINDEX);
    hipLaunchKernelGGL(k_bit_indices, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, keys, D, n_taxa,
                       split_dev, rows, cols);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// ---- zero-fill + scatter into the compact matrices ---------------------------------------------
// Zero the (R_pad x K_pad) region of every split's matrix with 16-byte stores (the Gram kernel
// reads whole 64 x 32 tiles, so the padding must be zero).  grid = (zblocks, S).
template <typename T>
__global__ __launch_bounds__(256) void k_zero(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                              T* __restrict__ mats) {
    const SplitDev& sp = splits[blockIdx.y];
    const int2 d = dims[blockIdx.y];
    const int rpad = min((d.x + 63) & ~63, sp.rcap);
    const int kpad = min((d.y + 31) & ~31, sp.pitch);
    constexpr int V = 16 / sizeof(T);  // elements per 16-byte store
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int kv = kpad / V;
    const int64_t total = (int64_t)rpad * kv;
    T* base = mats + sp.mat_off;
    vec_t z = {};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int row = (int)(e / kv), cv = (int)(e % kv);
        *reinterpret_cast<vec_t*>(base + (int64_t)row * sp.pitch + (int64_t)cv * V) = z;
    }
}

// grid = (ceil(D/256), S): M[s][rr][cc] = value  (assignment, constructions.py:43,:54,:101)
template <typename T>
__global__ __launch_bounds__(256) void k_scatter(const SplitDev* __restrict__ splits, int64_t D,
                                                 const u32* __restrict__ rr, const u32* __restrict__ cc,
                                                 const T* __restrict__ vals, T* __restrict__ mats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const SplitDev& sp = splits[blockIdx.y];
    const int64_t o = (int64_t)blockIdx.y * D + i;
    mats[sp.mat_off + (int64_t)rr[o] * sp.pitch + cc[o]] = vals[i];
}

template <typename T>
int launch_zero_scatter(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, int64_t D,
                        const int2* dims, const u32* rr, const u32* cc, const T* vals, T* mats) {
    if (splits.empty() || D == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_SCATTER);
    const int S = (int)splits.size();
    int64_t maxel = 0;
    for (const auto& s : splits) maxel = std::max<int64_t>(maxel, (int64_t)s.rcap * s.pitch);
    int zb = (int)std::min<int64_t>(64, std::max<int64_t>(1, maxel * sizeof(T) / 16 / 256 / 4));
    hipLaunchKernelGGL(k_zero<T>, dim3(zb, S), dim3(256), 0, ctx->stream, splits_dev, dims, mats);
    hipLaunchKernelGGL(k_scatter<T>, dim3((unsigned)((D + 255) / 256), S), dim3(256), 0, ctx->stream, splits_dev, D, rr,
                       cc, vals, mats);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
template int launch_zero_scatter<u32>(sp_ctx*, const SplitDev*, const std::vector<SplitDev>&, int64_t, const int2*,
                                      const u32*, const u32*, const u32*, u32*);
template int launch_zero_scatter<double>(sp_ctx*, const SplitDev*, const std::vector<SplitDev>&, int64_t, const int2*,
                                         const u32*, const u32*, const double*, double*);

// ---- the sorted used row / column keys themselves (for the reduced format's axes) --------------
__global__ void k_used_keys(const u64* __restrict__ keys, int64_t D, int n, const SplitDev* __restrict__ split,
                            const u32* __restrict__ rr, const u32* __restrict__ cc, int64_t* __restrict__ row_keys,
                            int64_t* __restrict__ col_keys) {
    __shared__ int shifts[32];
    const SplitDev& sp = split[0];
    load_shifts(shifts, sp, n);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    u64 r, c;
    split_rowcol(keys[i], shifts, sp.nr, sp.nc, r, c);
    row_keys[rr[i]] = (int64_t)r;  // all writers of one slot write the same value
    col_keys[cc[i]] = (int64_t)c;
}

int launch_used_keys(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* split_dev, const u32* rr,
                     const u32* cc, int64_t* row_keys, int64_t* col_keys) {
    if (D == 0) return SP_OK;
    hipLaunchKernelGGL(k_used_keys, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, keys, D, n_taxa,
                       split_dev, rr, cc, row_keys, col_keys);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// ---- full 4^a x 4^b count matrix ------------------------------------------------------------------
// One pass: every cell of the dense matrix is written exactly once.  A workgroup owns a slab of
// SLAB cells held in LDS: it zeroes the slab, scans the whole pattern table (D pairs, L2-resident,
// coalesced 8-byte reads), drops the counts that fall into its slab into LDS, and streams the slab
// out with 16-byte stores.  HBM traffic = 4 * 4^n bytes written + the table read once per XCD.
#define DENSE_SLAB 16384  // cells per workgroup (64 KiB of LDS)
__global__ __launch_bounds__(256) void k_dense_slab(const u64* __restrict__ keys, const u32* __restrict__ counts,
                                                    int64_t D, int n, const SplitDev* __restrict__ split,
                                                    u32* __restrict__ out, int64_t cells) {
    __shared__ __attribute__((aligned(16))) u32 slab[DENSE_SLAB];
    __shared__ int shifts[32];
    const SplitDev& sp = split[0];
    load_shifts(shifts, sp, n);
    const int nr = sp.nr, nc = sp.nc;
    const int64_t lo = (int64_t)blockIdx.x * DENSE_SLAB;
    const int64_t hi = min(cells, lo + DENSE_SLAB);
    for (int i = threadIdx.x; i < DENSE_SLAB; i += 256) slab[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < D; i += 256) {
        u64 r, c;
        split_rowcol(keys[i], shifts, nr, nc, r, c);
        const int64_t cell = (int64_t)((r << (2 * nc)) | c);
        if (cell >= lo && cell < hi) slab[cell - lo] = counts[i];
    }
    __syncthreads();
    const int nvec = (int)((hi - lo + 3) / 4);
    uint4* o = reinterpret_cast<uint4*>(out + lo);
    const uint4* s4 = reinterpret_cast<const uint4*>(slab);
    if (hi - lo == DENSE_SLAB || ((hi - lo) & 3) == 0) {
        for (int i = threadIdx.x; i < nvec; i += 256) o[i] = s4[i];
    } else {
        for (int64_t i = threadIdx.x; i < hi - lo; i += 256) out[lo + i] = slab[i];
    }
}

int launch_dense_scatter(sp_ctx* ctx, const u64* keys, const u32* counts, int64_t D, int n_taxa,
                         const SplitDev* split_dev, const SplitDev& split, u32* out) {
    PhaseScope ps(ctx, SP_PHASE_DENSE);
    const int64_t cells = pow4(split.nr + split.nc);
    const unsigned blocks = (unsigned)((cells + DENSE_SLAB - 1) / DENSE_SLAB);
    hipLaunchKernelGGL(k_dense_slab, dim3(blocks), dim3(256), 0, ctx->stream, keys, counts, D, n_taxa, split_dev, out,
                       cells);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// ---- int8 limb planes (input of gram_i8.hip) ------------------------------------------------------
// Split s owns nl planes of rcap x pitch bytes; plane l holds (count >> 7l) & 127.
__global__ __launch_bounds__(256) void k_zero_i8(const SplitDev* __restrict__ splits, const int2* __restrict__ dims,
                                                 uint8_t* __restrict__ mats, int nl) {
    const SplitDev& sp = splits[blockIdx.y];
    const int2 d = dims[blockIdx.y];
    const int rpad = min((d.x + 63) & ~63, sp.rcap);
    const int kpad = min((d.y + 127) & ~127, sp.pitch);
    const int kv = kpad / 16;
    const int64_t per_plane = (int64_t)rpad * kv;
    const int64_t total = per_plane * nl;
    const int64_t plane = (int64_t)sp.rcap * sp.pitch;
    uint8_t* base = mats + sp.mat_off;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int l = (int)(e / per_plane);
        const int64_t r = e % per_plane;
        const int row = (int)(r / kv), cv = (int)(r % kv);
        *reinterpret_cast<uint4*>(base + l * plane + (int64_t)row * sp.pitch + (int64_t)cv * 16) = z;
    }
}

__global__ __launch_bounds__(256) void k_scatter_i8(const SplitDev* __restrict__ splits, int64_t D,
                                                    const u32* __restrict__ rr, const u32* __restrict__ cc,
                                                    const u32* __restrict__ vals, uint8_t* __restrict__ mats, int nl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const SplitDev& sp = splits[blockIdx.y];
    const int64_t o = (int64_t)blockIdx.y * D + i;
    const int64_t plane = (int64_t)sp.rcap * sp.pitch;
    uint8_t* p = mats + sp.mat_off + (int64_t)rr[o] * sp.pitch + cc[o];
    u32 v = vals[i];
    for (int l = 0; l < nl; ++l) {   // (the planes were zero-filled: a zero limb - the high limb of most patterns - needs no store)
        if (v & 127u) p[l * plane] = (uint8_t)(v & 127u);
        v >>= 7;
    }
}

int launch_zero_scatter_i8(sp_ctx* ctx, int nl, const SplitDev* splits_dev, const std::vector<SplitDev>& splits,
                           int64_t D, const int2* dims, const u32* rr, const u32* cc, const u32* vals, uint8_t* mats) {
    if (splits.empty() || D == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_SCATTER);
    const int S = (int)splits.size();
    int64_t maxb = 0;
    for (const auto& s : splits) maxb = std::max<int64_t>(maxb, (int64_t)s.rcap * s.pitch * nl);
    const int zb = (int)std::min<int64_t>(64, std::max<int64_t>(1, maxb / 16 / 256 / 4));
    hipLaunchKernelGGL(k_zero_i8, dim3(zb, S), dim3(256), 0, ctx->stream, splits_dev, dims, mats, nl);
    hipLaunchKernelGGL(k_scatter_i8, dim3((unsigned)((D + 255) / 256), S), dim3(256), 0, ctx->stream, splits_dev, D, rr,
                       cc, vals, mats, nl);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
