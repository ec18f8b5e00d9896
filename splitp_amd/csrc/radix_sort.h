// Hand-written LSD radix sort + run-length encode for gfx950 (round 3; replaces rocprim::radix_sort_keys /
// run_length_encode in hist.hip and rocprim::segmented_radix_sort_pairs in sparse_big.hip).
//
// Stable least-significant-digit passes of RS_BITS = 8 bits.  One pass = three launches:
//   k_rs_hist     one workgroup (256 threads, 4 waves) per tile of RS_TILE = 4096 keys: digit histogram of the tile in LDS
//                 (wave-private rows, LDS atomics), written to hist[(segment * RADIX + digit) * blocks_per_segment + block]
//   k_rs_scan     exclusive scan of that array in exactly that order - segment-major, then digit, then tile: the offsets of
//                 an independent sort per segment (segments = equal-length contiguous pieces; 1 segment = a plain sort)
//   k_rs_scatter  the tile again: every key's rank among the keys of its digit inside the tile, by WAVE MATCH - eight
//                 ballots give a lane the mask of the lanes holding its digit, the rank inside the wave is a popcount below
//                 the lane, and the (item, wave) segments of the tile are chained through a table of per-segment digit
//                 counts in LDS (64 segments x 256 digits x u16 = 32 KB) scanned by digit - then out[offset + rank] = key
//                 (and value).  Ranks follow the tile order (item-major, then wave, then lane), so the pass is STABLE, which
//                 is what lets the passes compose into a sort; no atomics decide an order anywhere.
// Keys are read coalesced (item j of lane t of a tile = base + j * 256 + t).  HBM traffic per pass: 2 reads + 1 write of the
// key (+ value) arrays; a 1 M-key, 33-bit sort is 5 passes x 12 MB - launch-latency bound (15 launches), not bandwidth bound.
#pragma once
#include "common.h"

#define RS_BITS 8            // (9-bit digits - one pass fewer on 33-, 41- and 17-bit keys, a 64 KB rank table - measured the same at
#define RS_RADIX 256         // 1 M keys, 0.33 / 0.42 ms, and slower at 4 - 8 M keys, 0.65 / 1.03 against 0.40 / 0.86 ms: kept at 8)
#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)

template <typename KT>
__device__ __forceinline__ u32 rs_digit(KT key, int shift) { return (u32)((key >> shift) & (KT)(RS_RADIX - 1)); }

// hist[(seg * RADIX + d) * bps + blk]: keys of digit d in tile blk of segment seg
template <typename KT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const KT* __restrict__ keys, int64_t seg_len, int bps, int shift,
                                                        u32* __restrict__ hist, u32* __restrict__ tot) {
    __shared__ u32 cnt[RS_WAVES][RS_RADIX];
    const int seg = blockIdx.x / bps, blk = blockIdx.x % bps;
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)seg * seg_len, lo = (int64_t)blk * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j * RS_THREADS + threadIdx.x;
        if (i < seg_len) atomicAdd(&cnt[w][rs_digit(keys[base + i], shift)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RS_RADIX; d += RS_THREADS) {
        const u32 total = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
        hist[((size_t)seg * RS_RADIX + d) * bps + blk] = total;
        if (total) atomicAdd(&tot[seg * RS_RADIX + d], total);   // (integer: the order of the adds does not matter)
    }
}

// Offsets of one pass: workgroup (segment, digit) turns its row hist[(seg * RADIX + d) * bps + 0 .. bps) - contiguous, read
// coalesced - into exclusive prefixes and adds the row's base = seg * seg_len + (keys of smaller digits in the segment, from
// the per-digit totals k_rs_hist accumulated).  RADIX x segments workgroups instead of one (a single-workgroup scan of the
// 500 k counters of an 8 M-key sort took 0.5 ms per pass).  Also zeroes the totals of the NEXT pass (the other buffer).
static __global__ __launch_bounds__(RS_THREADS) void k_rs_offsets(u32* __restrict__ hist, const u32* __restrict__ tot,
                                                           u32* __restrict__ tot_next, int64_t seg_len, int bps) {
    __shared__ u32 wsum[RS_WAVES + 1];
    const int seg = blockIdx.x / RS_RADIX, d = blockIdx.x % RS_RADIX;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 below = 0;
    for (int t = threadIdx.x; t < d; t += RS_THREADS) below += tot[seg * RS_RADIX + t];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) below += __shfl_xor(below, k, 64);
    if (lane == 0) wsum[w] = below;
    __syncthreads();
    u32 run = (u32)((int64_t)seg * seg_len) + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    u32* row = hist + ((size_t)seg * RS_RADIX + d) * bps;
    for (int b0 = 0; b0 < bps; b0 += RS_THREADS) {
        const int b = b0 + (int)threadIdx.x;
        const u32 v = b < bps ? row[b] : 0u;
        u32 x = v;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const u32 y = __shfl_up(x, k, 64);
            if (lane >= k) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        u32 before = 0, all = 0;
#pragma unroll
        for (int i = 0; i < RS_WAVES; ++i) {
            if (i < w) before += wsum[i];
            all += wsum[i];
        }
        if (b < bps) row[b] = run + before + x - v;
        run += all;
        __syncthreads();
    }
    if (d == 0)
        for (int t = threadIdx.x; t < RS_RADIX; t += RS_THREADS) tot_next[seg * RS_RADIX + t] = 0;
}

// in-place exclusive scan of n u32 by ONE workgroup of 1024 threads (n = segments x 256 x tiles: 63 k words for a
// 1 M-key sort; every thread scans a contiguous piece, the pieces are chained through one block scan)
static __global__ __launch_bounds__(1024) void k_rs_scan(u32* __restrict__ a, int64_t n) {
    __shared__ unsigned long long wsum[16];
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo0 = (int64_t)threadIdx.x * per, lo = lo0 < n ? lo0 : n, hi = lo + per < n ? lo + per : n;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; ++i) s += a[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long x = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    unsigned long long before = 0;
    for (int i = 0; i < w; ++i) before += wsum[i];
    unsigned long long run = before + x - s;
    for (int64_t i = lo; i < hi; ++i) {
        const u32 v = a[i];
        a[i] = (u32)run;
        run += v;
    }
}

template <typename KT, bool VALUES>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const KT* __restrict__ keys, const u32* __restrict__ vals,
                                                           int64_t seg_len, int bps, int shift,
                                                           const u32* __restrict__ offs, KT* __restrict__ out_keys,
                                                           u32* __restrict__ out_vals) {
    // count of every digit in every (item, wave) segment of the tile, then - in place - its exclusive prefix over the segments
    __shared__ unsigned short segcnt[RS_ITEMS * RS_WAVES][RS_RADIX];
    const int seg = blockIdx.x / bps, blk = blockIdx.x % bps;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_ITEMS * RS_WAVES * RS_RADIX / 2; i += RS_THREADS) reinterpret_cast<u32*>(&segcnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)seg * seg_len, lo = (int64_t)blk * RS_TILE;
    KT key[RS_ITEMS];
    u32 val[RS_ITEMS];
    unsigned short rank[RS_ITEMS];
    unsigned short dig[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j * RS_THREADS + threadIdx.x;
        const bool in = i < seg_len;
        key[j] = in ? keys[base + i] : (KT)0;
        if (VALUES) val[j] = in ? vals[base + i] : 0u;
        const u32 d = rs_digit(key[j], shift);
        // lanes of this wave with the same digit (and inside the array)
        unsigned long long m = __ballot(in);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        dig[j] = (unsigned short)d;
        rank[j] = (unsigned short)__popcll(m & ((1ull << lane) - 1ull));
        if (in && rank[j] == 0) segcnt[j * RS_WAVES + w][d] = (unsigned short)__popcll(m);   // (the first lane of the group)
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RS_RADIX; d += RS_THREADS) {   // exclusive prefix over the 64 segments, digit by digit
        unsigned int run = 0;                                       // (tile order = item-major, then wave)
#pragma unroll 8
        for (int s = 0; s < RS_ITEMS * RS_WAVES; ++s) {
            const unsigned int c = segcnt[s][d];
            segcnt[s][d] = (unsigned short)run;
            run += c;
        }
    }
    __syncthreads();
    const u32* __restrict__ my_offs = offs + (size_t)seg * RS_RADIX * bps + blk;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j * RS_THREADS + threadIdx.x;
        if (i < seg_len) {
            const u32 d = dig[j];
            // (offs are positions inside the whole array: the scan runs over all segments)
            const size_t pos = (size_t)my_offs[(size_t)d * bps] + segcnt[j * RS_WAVES + w][d] + rank[j];
            out_keys[pos] = key[j];
            if (VALUES) out_vals[pos] = val[j];
        }
    }
}

// Stable sort of `n_seg` independent segments of `seg_len` keys each (n_seg = 1: one array) on the bits [0, end_bit).
// keys_in (and vals_in, optional values carried with the keys) are only read; the passes ping-pong between the two work
// arrays keys_a / keys_b (vals_a / vals_b), n = seg_len * n_seg elements each, and the result ends up in *sorted_keys
// (*sorted_vals), one of the two.  work: n_seg * RADIX * (tiles + 2) + 16 u32 (grown here).
template <typename KT>
static int rs_sort(sp_ctx* ctx, const KT* keys_in, KT* keys_a, KT* keys_b, const u32* vals_in, u32* vals_a, u32* vals_b,
                   int64_t seg_len, int64_t n_seg, unsigned end_bit, DevBuf& work, const KT** sorted_keys,
                   const u32** sorted_vals) {
    *sorted_keys = keys_in;
    if (sorted_vals) *sorted_vals = vals_in;
    if (seg_len <= 0 || n_seg <= 0 || end_bit == 0) return SP_OK;
    const int64_t bps = (seg_len + RS_TILE - 1) / RS_TILE;
    const int64_t blocks = bps * n_seg;
    SP_REQUIRE(blocks < ((int64_t)1 << 31) && seg_len * n_seg < ((int64_t)1 << 32), SP_ELIMIT,
               "radix sort: %lld keys in %lld segments (limit 2^32 keys)", (long long)(seg_len * n_seg), (long long)n_seg);
    const int64_t nh = blocks * RS_RADIX;
    SP_CHECK(work.ensure((size_t)(nh + 2 * n_seg * RS_RADIX + 16) * 4));
    u32* hist = work.as<u32>();
    u32* tot2 = hist + nh;                                    // two buffers of per-(segment, digit) totals, used alternately
    SP_HIP(hipMemsetAsync(tot2, 0, (size_t)2 * n_seg * RS_RADIX * 4, ctx->stream));
    int pass = 0;
    const KT* src = keys_in;
    const u32* vsrc = vals_in;
    KT* dst = keys_a;
    u32* vdst = vals_a;
    for (unsigned shift = 0; shift < end_bit; shift += RS_BITS) {
        u32* tot = tot2 + (size_t)(pass & 1) * n_seg * RS_RADIX;
        u32* tot_next = tot2 + (size_t)((pass + 1) & 1) * n_seg * RS_RADIX;
        ++pass;
        hipLaunchKernelGGL(k_rs_hist<KT>, dim3((unsigned)blocks), dim3(RS_THREADS), 0, ctx->stream, src, seg_len, (int)bps,
                           (int)shift, hist, tot);
        hipLaunchKernelGGL(k_rs_offsets, dim3((unsigned)(n_seg * RS_RADIX)), dim3(RS_THREADS), 0, ctx->stream, hist,
                           (const u32*)tot, tot_next, seg_len, (int)bps);
        if (vals_in)
            hipLaunchKernelGGL((k_rs_scatter<KT, true>), dim3((unsigned)blocks), dim3(RS_THREADS), 0, ctx->stream, src, vsrc,
                               seg_len, (int)bps, (int)shift, (const u32*)hist, dst, vdst);
        else
            hipLaunchKernelGGL((k_rs_scatter<KT, false>), dim3((unsigned)blocks), dim3(RS_THREADS), 0, ctx->stream, src,
                               (const u32*)nullptr, seg_len, (int)bps, (int)shift, (const u32*)hist, dst, (u32*)nullptr);
        SP_HIP(hipGetLastError());
        src = dst;
        vsrc = vdst;
        dst = dst == keys_a ? keys_b : keys_a;
        vdst = vdst == vals_a ? vals_b : vals_a;
    }
    *sorted_keys = src;
    if (sorted_vals) *sorted_vals = vsrc;
    return SP_OK;
}

// ---- run-length encode of a sorted array ---------------------------------------------------------------------------------
// heads[b] = run heads in tile b (a head: key differs from its predecessor; element 0 is one); after the scan, tile b's first
// head is run heads[b].  k_rle_write stores the key and the start position of every run, k_rle_counts turns consecutive
// starts into lengths.  n_runs_out[0] = number of runs.
template <typename KT>
__global__ __launch_bounds__(RS_THREADS) void k_rle_heads(const KT* __restrict__ s, int64_t n, u32* __restrict__ heads) {
    __shared__ u32 wcnt[RS_WAVES];
    const int64_t lo = (int64_t)blockIdx.x * RS_TILE;
    u32 c = 0;
#pragma unroll 4
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j * RS_THREADS + threadIdx.x;
        if (i < n) c += (i == 0 || s[i] != s[i - 1]) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) heads[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

template <typename KT>
__global__ __launch_bounds__(RS_THREADS) void k_rle_write(const KT* __restrict__ s, int64_t n, const u32* __restrict__ heads,
                                                          int64_t n_tiles, KT* __restrict__ uniq, u32* __restrict__ starts,
                                                          u32* __restrict__ n_runs_out) {
    // tile order here is CONTIGUOUS per thread (thread t owns elements lo + t * ITEMS ..): run numbers follow array order
    __shared__ u32 wsum[RS_WAVES];
    const int64_t lo = (int64_t)blockIdx.x * RS_TILE + (int64_t)threadIdx.x * RS_ITEMS;
    bool head[RS_ITEMS];
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j;
        head[j] = i < n && (i == 0 || s[i] != s[i - 1]);
        c += head[j] ? 1u : 0u;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 x = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    u32 before = heads[blockIdx.x];
    for (int i = 0; i < w; ++i) before += wsum[i];
    u32 run = before + x - c;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j)
        if (head[j]) {
            uniq[run] = s[lo + j];
            starts[run] = (u32)(lo + j);
            ++run;
        }
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == RS_THREADS - 1) {
        n_runs_out[0] = run;
        starts[run] = (u32)n;      // sentinel: the end of the last run
    }
}

static __global__ void k_rle_counts(const u32* __restrict__ starts, const u32* __restrict__ n_runs, u32* __restrict__ counts) {
    const u32 nr = n_runs[0];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) counts[i] = starts[i + 1] - starts[i];
}

// uniq[r], counts[r] for the runs of the sorted array s[0 .. n); n_runs_dev[0] = number of runs (device word).
// work: (tiles + n + 32) u32.  n < 2^32.
template <typename KT>
static int rs_run_length_encode(sp_ctx* ctx, const KT* s, int64_t n, KT* uniq, u32* counts, u32* n_runs_dev, DevBuf& work) {
    if (n <= 0) {
        SP_HIP(hipMemsetAsync(n_runs_dev, 0, 4, ctx->stream));
        return SP_OK;
    }
    const int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
    SP_CHECK(work.ensure((size_t)(tiles + n + 32) * 4));
    u32* heads = work.as<u32>();
    u32* starts = heads + tiles + 8;
    hipLaunchKernelGGL(k_rle_heads<KT>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, ctx->stream, s, n, heads);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, ctx->stream, heads, tiles);
    hipLaunchKernelGGL(k_rle_write<KT>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, ctx->stream, s, n, (const u32*)heads, tiles, uniq,
                       starts, n_runs_dev);
    hipLaunchKernelGGL(k_rle_counts, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const u32*)starts, (const u32*)n_runs_dev, counts);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
