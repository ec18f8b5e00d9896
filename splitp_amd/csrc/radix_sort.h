// Hand-written LSD radix sort + run-length encode for gfx950 (round 3; replaces rocprim::radix_sort_keys /
// run_length_encode in hist.hip and rocprim::segmented_radix_sort_pairs in sparse_big.hip).
//
// Stable least-significant-digit passes of RS_BITS = 8 bits over `segments` equal-length contiguous pieces (1 segment = a
// plain sort).  ONE-SWEEP form (round 4, rs_sort): 2 + passes launches instead of 3 x passes.
//   memset        digit totals, tile tickets and the look-back words of all passes
//   k_os_hist     every key read once: the digit totals of EVERY pass per segment (the multiset of a segment's keys does
//                 not change from pass to pass), LDS histogram per workgroup of RS_HTILES tiles, flushed by global atomics
//                 (integers: any order)
//   k_os_pass     one launch per pass, one workgroup (256 threads, 4 waves) per tile of RS_TILE = 4096 keys.  A tile takes
//                 its number from a ticket counter (so every tile with a smaller number is already running or done), ranks
//                 its keys by WAVE MATCH - eight ballots give a lane the mask of the lanes holding its digit, the rank inside
//                 the wave is a popcount below the lane, and the 64 (item, wave) pieces of the tile are chained through a
//                 table of per-piece digit counts in LDS (64 x 256 x u16 = 32 KB) scanned by digit - and learns where its
//                 digits start from the tiles before it in the segment by DECOUPLED LOOK-BACK: thread d publishes
//                 (AGGREGATE | count of digit d in this tile) in the tile's look-back word, walks back over the
//                 predecessors' words (spinning on a word that is still empty) adding their aggregates until it meets an
//                 INCLUSIVE PREFIX, and publishes its own inclusive prefix.  Flag and value share one 32-bit word written
//                 and read with device-scope atomics: no fence, nothing is ever read that was not published whole.
//                 The first tile of a segment publishes a prefix at once, so a chain never leaves its segment.
//   Ranks follow the tile order (item-major, then wave, then lane) and tiles the array order, so every pass is STABLE - which
//   is what lets the passes compose into a sort; no atomic decides an order anywhere (the ticket only names the tile).
// Keys are read coalesced (item j of lane t of a tile = base + j * 256 + t).  HBM traffic: one read for the totals, then one
// read + one write of the key (+ value) arrays per pass.
// The three-launch form of round 3 (k_rs_hist: tile histograms -> k_rs_offsets: one workgroup per (segment, digit) row turns
// the tile counts into offsets, per-digit totals accumulated atomically in one of two alternating buffers, the d == 0 block of
// pass p zeroing the buffer pass p + 1 adds into -> k_rs_scatter) stays selectable (context option `sort_three_launch`) as the
// cross-check of the look-back form.  k_rs_scan serves the run-length encode only.
#pragma once
#include "common.h"

#define RS_BITS 8            // (9-bit digits - one pass fewer on 33-, 41- and 17-bit keys, a 64 KB rank table - measured the same at
#define RS_RADIX 256         // 1 M keys, 0.33 / 0.42 ms, and slower at 4 - 8 M keys, 0.65 / 1.03 against 0.40 / 0.86 ms: kept at 8)
#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_HTILES 1          // tiles per workgroup of the one-sweep histogram kernel (8: 31 workgroups for 1 M keys, 180 us)
#define OS_LB 8              // predecessors whose look-back words a thread requests together
#define OS_AGG 0x40000000u   // look-back word: flag in the two top bits (0 = not published yet), 30-bit count below
#define OS_PREF 0x80000000u
#define OS_MASK 0x3FFFFFFFu

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

// ---- one-sweep form -------------------------------------------------------------------------------------------------------
// Digit width BITS = 8 (default) or 9 (option `sort_digit_bits`: one pass fewer on 17-, 33- and 41-bit keys, slower per pass -
// see rs_sort_nine_bit_digits).  RADIX = 2^BITS; a thread owns RADIX / 256 digits.
template <typename KT, int BITS>
__device__ __forceinline__ u32 os_digit(KT key, int shift) { return (u32)((key >> shift) & (KT)((1u << BITS) - 1u)); }

// ghist[(seg * n_pass + p) * RADIX + d] += keys of segment seg whose digit p is d.  Grid = n_seg * bph workgroups.
// The upper digits of a site-pattern key take few values (33-bit keys: the last digit has two) - as one LDS atomic per key
// and pass on a shared row the kernel spent 70 us on 1 M keys, 256 threads queueing on a handful of counters.  So the lanes
// of a wave that hold the same digit are matched first (BITS ballots, as in the pass kernel) and the first lane of each
// group adds the group's size to a wave-private row: one atomic per distinct digit, wave and pass.
template <typename KT, int BITS>
__global__ __launch_bounds__(RS_THREADS) void k_os_hist(const KT* __restrict__ keys, int64_t seg_len, int bph, int n_pass,
                                                        u32* __restrict__ ghist) {
    constexpr int RADIX = 1 << BITS;
    constexpr int MAXP = (int)(8 * sizeof(KT) + BITS - 1) / BITS;   // passes a key of this width can need
    __shared__ u32 cnt[RS_WAVES][MAXP][RADIX];   // <= 64 KB
    const int seg = blockIdx.x / bph, blk = blockIdx.x % bph;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * MAXP * RADIX; i += RS_THREADS) (&cnt[0][0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)seg * seg_len, lo = (int64_t)blk * RS_TILE * RS_HTILES;
    const int64_t hi = lo + (int64_t)RS_TILE * RS_HTILES < seg_len ? lo + (int64_t)RS_TILE * RS_HTILES : seg_len;
    for (int64_t t0 = lo; t0 < hi; t0 += RS_TILE) {   // (uniform trip counts: the ballots need whole waves)
        KT kk[RS_ITEMS];
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {   // all of the tile's loads in flight before the first ballot
            const int64_t i = t0 + j * RS_THREADS + threadIdx.x;
            kk[j] = i < hi ? keys[base + i] : (KT)0;
        }
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const bool in = t0 + j * RS_THREADS + threadIdx.x < hi;
            const KT k = kk[j];
            const unsigned long long inm = __ballot(in);
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                if (p < n_pass) {
                    const u32 d = os_digit<KT, BITS>(k, p * BITS);
                    unsigned long long m = inm;
#pragma unroll
                    for (int b = 0; b < BITS; ++b) {
                        const unsigned long long bal = __ballot((d >> b) & 1u);
                        m &= ((d >> b) & 1u) ? bal : ~bal;
                    }
                    if (in && (m & ((1ull << lane) - 1ull)) == 0) atomicAdd(&cnt[w][p][d], (u32)__popcll(m));
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_pass * RADIX; e += RS_THREADS) {
        u32 c = 0;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) c += (&cnt[ww][0][0])[e];
        if (c) atomicAdd(&ghist[(size_t)seg * n_pass * RADIX + e], c);
    }
}

// One pass.  lookback: (tiles of all segments) x RADIX words of THIS pass, all zero at launch; ticket: this pass's counter.
template <typename KT, bool VALUES, int BITS>
__global__ __launch_bounds__(RS_THREADS) void k_os_pass(const KT* __restrict__ keys, const u32* __restrict__ vals,
                                                        int64_t seg_len, int bps, int shift, int pass, int n_pass,
                                                        const u32* __restrict__ ghist, u32* __restrict__ lookback,
                                                        u32* __restrict__ ticket, KT* __restrict__ out_keys,
                                                        u32* __restrict__ out_vals) {
    constexpr int RADIX = 1 << BITS;
    constexpr int NDIG = RADIX / RS_THREADS;   // digits a thread owns: d = threadIdx.x + q * RS_THREADS
    __shared__ unsigned short segcnt[RS_ITEMS * RS_WAVES][RADIX];   // 32 / 64 KB
    __shared__ u32 dbase[RADIX];
    __shared__ u32 wsum[NDIG][RS_WAVES];
    __shared__ u32 s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < RS_ITEMS * RS_WAVES * RADIX / 2; i += RS_THREADS) reinterpret_cast<u32*>(&segcnt[0][0])[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    const int seg = (int)(tile / (u32)bps), blk = (int)(tile % (u32)bps);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
        const u32 d = os_digit<KT, BITS>(key[j], shift);
        unsigned long long m = __ballot(in);   // lanes of this wave with the same digit (and inside the array)
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        dig[j] = (unsigned short)d;
        rank[j] = (unsigned short)__popcll(m & ((1ull << lane) - 1ull));
        if (in && rank[j] == 0) segcnt[j * RS_WAVES + w][d] = (unsigned short)__popcll(m);   // (the first lane of the group)
    }
    __syncthreads();
    // per owned digit: exclusive prefix over the 64 pieces (tile order = item-major, then wave); its total = the tile's count
    u32 mine[NDIG], tot_d[NDIG], x[NDIG];
#pragma unroll
    for (int q = 0; q < NDIG; ++q) {
        const int d = threadIdx.x + q * RS_THREADS;
        unsigned int run = 0;
#pragma unroll 8
        for (int sgi = 0; sgi < RS_ITEMS * RS_WAVES; ++sgi) {
            const unsigned int c = segcnt[sgi][d];
            segcnt[sgi][d] = (unsigned short)run;
            run += c;
        }
        mine[q] = run;
        // where digit d starts in the segment: keys of smaller digits (exclusive scan of this pass's totals over the digits)
        tot_d[q] = ghist[((size_t)seg * n_pass + pass) * RADIX + d];
        u32 xx = tot_d[q];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const u32 y = __shfl_up(xx, k, 64);
            if (lane >= k) xx += y;
        }
        x[q] = xx;
        if (lane == 63) wsum[q][w] = xx;
    }
    __syncthreads();
    u32 carry = 0;   // totals of the digit blocks before block q
#pragma unroll
    for (int q = 0; q < NDIG; ++q) {
        const int d = threadIdx.x + q * RS_THREADS;
        u32 below = carry + x[q] - tot_d[q];
        u32 blocksum = 0;
#pragma unroll
        for (int i = 0; i < RS_WAVES; ++i) {
            if (i < w) below += wsum[q][i];
            blocksum += wsum[q][i];
        }
        carry += blocksum;
        // decoupled look-back over the earlier tiles of this segment
        u32* const lb = lookback + (size_t)tile * RADIX + d;
        u32 excl = 0;
        if (blk == 0) {
            __hip_atomic_store(lb, OS_PREF | mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(lb, OS_AGG | mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // OS_LB predecessors per round, their words requested together: when all tiles of a pass start at once no tile
            // but the first holds a prefix yet and tile t walks back ~t / 2 words - one dependent L2 round trip each otherwise
            const u32* p = lb - RADIX;
            bool done = false;
            for (int back = 0; back < blk && !done; back += OS_LB, p -= OS_LB * RADIX) {   // (ends at the segment's first tile)
                u32 v[OS_LB];
#pragma unroll
                for (int u = 0; u < OS_LB; ++u)
                    v[u] = back + u < blk ? __hip_atomic_load(p - u * RADIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_PREF;
#pragma unroll
                for (int u = 0; u < OS_LB; ++u) {
                    if (!done) {
                        u32 xv = v[u];
                        int spins = 0;
                        // the tile holds a smaller ticket: it is running and publishes its aggregate without waiting for
                        // anybody.  (The spin is bounded all the same - ~50 ms - so that a defect could only ever produce a
                        // wrong order, which the tests catch, never a wave that does not finish.)
                        while ((xv >> 30) == 0u && ++spins < (1 << 20)) {
                            __builtin_amdgcn_s_sleep(1);
                            xv = __hip_atomic_load(p - u * RADIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        excl += xv & OS_MASK;
                        if (xv & OS_PREF) done = true;
                    }
                }
            }
            __hip_atomic_store(lb, OS_PREF | (excl + mine[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dbase[d] = (u32)((int64_t)seg * seg_len) + below + excl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = lo + j * RS_THREADS + threadIdx.x;
        if (i < seg_len) {
            const u32 dd = dig[j];
            const size_t pos = (size_t)dbase[dd] + segcnt[j * RS_WAVES + w][dd] + rank[j];
            out_keys[pos] = key[j];
            if (VALUES) out_vals[pos] = val[j];
        }
    }
}

template <typename KT>
static int rs_sort_three_launch(sp_ctx* ctx, const KT* keys_in, KT* keys_a, KT* keys_b, const u32* vals_in, u32* vals_a, u32* vals_b,
                                int64_t seg_len, int64_t n_seg, unsigned end_bit, DevBuf& work, const KT** sorted_keys,
                                const u32** sorted_vals);

template <typename KT, int BITS>
static int rs_sort_onesweep(sp_ctx* ctx, const KT* keys_in, KT* keys_a, KT* keys_b, const u32* vals_in, u32* vals_a, u32* vals_b,
                            int64_t seg_len, int64_t n_seg, unsigned end_bit, DevBuf& work, const KT** sorted_keys,
                            const u32** sorted_vals) {
    constexpr int RADIX = 1 << BITS;
    const int n_pass = (int)((end_bit + BITS - 1) / BITS);
    // (tiles of 2048 keys for sorts with fewer than two 4096-key tiles per CU measured no faster: 25 against 23 us a pass at 1 M keys)
    const int64_t bps = (seg_len + RS_TILE - 1) / RS_TILE;
    const int64_t blocks = bps * n_seg;
    SP_REQUIRE(blocks < ((int64_t)1 << 31) && seg_len * n_seg < ((int64_t)1 << 32) && seg_len < ((int64_t)1 << 30) &&
                   end_bit <= 8 * sizeof(KT),
               SP_ELIMIT, "radix sort: %lld keys in %lld segments, %u key bits (limits: 2^32 keys, 2^30 per segment, the key's width)",
               (long long)(seg_len * n_seg), (long long)n_seg, end_bit);
    const size_t n_tot = (size_t)n_seg * n_pass * RADIX, n_lb = (size_t)blocks * RADIX;
    const size_t words = n_tot + 16 + (size_t)n_pass * n_lb;
    SP_CHECK(work.ensure(words * 4));
    u32* ghist = work.as<u32>();
    u32* tickets = ghist + n_tot;
    u32* lookback = tickets + 16;
    SP_HIP(hipMemsetAsync(ghist, 0, words * 4, ctx->stream));
    const int64_t bph = (seg_len + (int64_t)RS_TILE * RS_HTILES - 1) / ((int64_t)RS_TILE * RS_HTILES);
    hipLaunchKernelGGL((k_os_hist<KT, BITS>), dim3((unsigned)(bph * n_seg)), dim3(RS_THREADS), 0, ctx->stream, keys_in, seg_len,
                       (int)bph, n_pass, ghist);
    const KT* src = keys_in;
    const u32* vsrc = vals_in;
    KT* dst = keys_a;
    u32* vdst = vals_a;
    for (int pass = 0; pass < n_pass; ++pass) {
        u32* lb = lookback + (size_t)pass * n_lb;
        const dim3 grid((unsigned)blocks), blk(RS_THREADS);
        if (vals_in)
            hipLaunchKernelGGL((k_os_pass<KT, true, BITS>), grid, blk, 0, ctx->stream, src, vsrc, seg_len, (int)bps, pass * BITS,
                               pass, n_pass, (const u32*)ghist, lb, tickets + pass, dst, vdst);
        else
            hipLaunchKernelGGL((k_os_pass<KT, false, BITS>), grid, blk, 0, ctx->stream, src, (const u32*)nullptr, seg_len, (int)bps,
                               pass * BITS, pass, n_pass, (const u32*)ghist, lb, tickets + pass, dst, (u32*)nullptr);
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

// Number of passes rs_sort makes for keys of end_bit bits (callers that need the result in a particular one of the two work
// arrays pick the first target by its parity): 9-bit digits where they save a pass, unless an option pins the width.
// (Measured, round 4: 9-bit digits on the 17-bit compact ids of the big-table form - 2 passes instead of 3 - ran 1.20 ms a
// pass against 0.69 ms: the 64 KB rank table halves the workgroups per CU and doubles the per-digit scan; 36 x 1.20 = 43 ms
// against 54 x 0.69 = 37 ms per three calls.  8 bits stay the default; `sort_digit_bits` = 9 keeps the wider form under test.)
static inline bool rs_sort_nine_bit_digits(const sp_ctx* ctx, unsigned end_bit) {
    (void)end_bit;
    return !ctx->opt.sort_three_launch && ctx->opt.sort_digit_bits == 9;
}
static inline unsigned rs_sort_passes(const sp_ctx* ctx, unsigned end_bit) {
    const unsigned b = rs_sort_nine_bit_digits(ctx, end_bit) ? 9u : 8u;
    return (end_bit + b - 1) / b;
}

// Stable sort of `n_seg` independent segments of `seg_len` keys each (n_seg = 1: one array) on the bits [0, end_bit).
// keys_in (and vals_in, optional values carried with the keys) are only read; the passes ping-pong between the two work
// arrays keys_a / keys_b (vals_a / vals_b), n = seg_len * n_seg elements each, and the result ends up in *sorted_keys
// (*sorted_vals), one of the two.  work: grown here (digit totals + tickets + one look-back word per tile, digit and pass).
template <typename KT>
static int rs_sort(sp_ctx* ctx, const KT* keys_in, KT* keys_a, KT* keys_b, const u32* vals_in, u32* vals_a, u32* vals_b,
                   int64_t seg_len, int64_t n_seg, unsigned end_bit, DevBuf& work, const KT** sorted_keys,
                   const u32** sorted_vals) {
    if (ctx->opt.sort_three_launch)
        return rs_sort_three_launch<KT>(ctx, keys_in, keys_a, keys_b, vals_in, vals_a, vals_b, seg_len, n_seg, end_bit, work,
                                        sorted_keys, sorted_vals);
    *sorted_keys = keys_in;
    if (sorted_vals) *sorted_vals = vals_in;
    if (seg_len <= 0 || n_seg <= 0 || end_bit == 0) return SP_OK;
    if (rs_sort_nine_bit_digits(ctx, end_bit))
        return rs_sort_onesweep<KT, 9>(ctx, keys_in, keys_a, keys_b, vals_in, vals_a, vals_b, seg_len, n_seg, end_bit, work, sorted_keys,
                                       sorted_vals);
    return rs_sort_onesweep<KT, 8>(ctx, keys_in, keys_a, keys_b, vals_in, vals_a, vals_b, seg_len, n_seg, end_bit, work, sorted_keys,
                                   sorted_vals);
}

// The three-launch form of round 3 (cross-check of the one-sweep form: context option `sort_three_launch`).
template <typename KT>
static int rs_sort_three_launch(sp_ctx* ctx, const KT* keys_in, KT* keys_a, KT* keys_b, const u32* vals_in, u32* vals_a, u32* vals_b,
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
// counts == nullptr: the run lengths are left to the caller, who finds the run starts (+ the sentinel n) at *starts_out.
template <typename KT>
static int rs_run_length_encode(sp_ctx* ctx, const KT* s, int64_t n, KT* uniq, u32* counts, u32* n_runs_dev, DevBuf& work,
                                const u32** starts_out = nullptr) {
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
    if (starts_out) *starts_out = starts;
    if (counts)
        hipLaunchKernelGGL(k_rle_counts, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u32*)starts, (const u32*)n_runs_dev, counts);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
