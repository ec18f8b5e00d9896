// Site-pattern histogram: alignment columns -> (pattern key, count) table, on the device.
//
// The step that immediately precedes the path in the reference:
//   splitp/parsers/fasta.py:48-63  get_pattern_counts (per-site Python loop; sites with any character
//                                  outside ACGT after .upper() are skipped and do not count towards N)
//   splitp/simulation.py:42-56     generate_alignment's counts dict
//
// Kernels
//   k_pack_sites   n ASCII rows of L characters -> one 2-bit-per-taxon key per site (coalesced row reads),
//                  invalid sites get the sentinel ~0.
//   k_hist_lds     the histogram proper.  Keys stream in coalesced 8-byte reads; each workgroup first
//                  aggregates its tile of sites in an LDS open-addressing table (ds_cmpst + ds_add), so the
//                  very hot bins (at short branch lengths the four constant patterns hold ~10 % of the
//                  sites EACH) cost one global atomic per workgroup instead of one per site; the table is
//                  then flushed to the 4^n-bin count array in HBM with global_atomic_add.
//   k_bins_count / k_bins_write   stream the bin array once each: per-block non-zero counts, exclusive scan
//                  on the host-free path (single-block scan kernel), ordered compaction -> keys ascending
//                  (= the A<C<G<T pattern order of simulation.py:51-54).
// The bin array is 4 * 4^n bytes: 4 MiB at n = 10, 64 MiB at n = 12, 16 GiB at n = 16 - affordable on a
// 288 GB part, which is why no sort is needed up to 16 taxa.
#include <algorithm>
#include <cstring>

#include "common.h"

#include "radix_sort.h"   // hand-written LSD radix sort + run-length encode for the sort-based histogram

#define HIST_THREADS 256
#define HIST_PER_THREAD 8
#define HIST_TILE (HIST_THREADS * HIST_PER_THREAD)  // sites per workgroup pass
#define HIST_SLOTS 8192                             // LDS table slots (4 * HIST_TILE): 64-96 KiB

template <typename KT>
__global__ __launch_bounds__(256) void k_pack_sites(const uint8_t* __restrict__ seqs, int n, int64_t L, int64_t stride,
                                                    KT* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    u64 k = 0;
    bool ok = true;
    for (int t = 0; t < n; ++t) {
        const unsigned ch = seqs[(int64_t)t * stride + i] & 0xDFu;  // upper-case (fasta.py:53)
        unsigned d;
        if (ch == 'A') d = 0;
        else if (ch == 'C') d = 1;
        else if (ch == 'G') d = 2;
        else if (ch == 'T') d = 3;
        else { d = 0; ok = false; }
        // '&0xDF' maps a few non-letters onto letters (e.g. '!' -> 0x01: no; 'a'-'z' -> 'A'-'Z' only for
        // 0x61-0x7A; characters 0x41-0x5A are unchanged).  Bytes 0x01-0x1A / 0x21-0x3A cannot alias A,C,G,T.
        k = (k << 2) | d;
    }
    keys[i] = ok ? (KT)k : (KT)~(KT)0;
}

// KT = u32 (n <= 15: half the HBM traffic of the pass) or u64; the all-ones key is the "invalid site" sentinel.
template <typename KT>
__global__ __launch_bounds__(HIST_THREADS) void k_hist_lds(const KT* __restrict__ keys, int64_t L, int n,
                                                           u32* __restrict__ bins, u32* __restrict__ n_valid) {
    __shared__ KT t_key[HIST_SLOTS];
    __shared__ u32 t_cnt[HIST_SLOTS];
    __shared__ u32 n_used;
    const KT SENT = (KT)~(KT)0;
    // the four constant patterns AAAA.., CCCC.., GGGG.., TTTT..: at short branch lengths each holds ~10 % of the sites.
    // They are counted per wave with a ballot + popcount (no LDS traffic at all) and flushed with one atomic per wave.
    const KT c1 = (KT)((n >= 32 ? ~0ull : ((1ull << (2 * n)) - 1)) / 3ull);
    u32 cc0 = 0, cc1 = 0, cc2 = 0, cc3 = 0, valid = 0;
    for (int i = threadIdx.x; i < HIST_SLOTS; i += HIST_THREADS) {
        t_key[i] = SENT;
        t_cnt[i] = 0;
    }
    if (threadIdx.x == 0) n_used = 0;
    __syncthreads();
    for (int64_t tile = (int64_t)blockIdx.x * HIST_TILE; tile < L; tile += (int64_t)gridDim.x * HIST_TILE) {
        KT kk[HIST_PER_THREAD];
#pragma unroll
        for (int j = 0; j < HIST_PER_THREAD; ++j) {       // all loads first (coalesced), then the LDS work
            const int64_t i = tile + (int64_t)j * HIST_THREADS + threadIdx.x;
            kk[j] = i < L ? keys[i] : SENT;
        }
#pragma unroll
        for (int j = 0; j < HIST_PER_THREAD; ++j) {
            const KT k = kk[j];
            const bool ok = k != SENT;
            const bool is0 = ok && k == (KT)0, is1 = ok && k == c1, is2 = ok && k == (KT)(2 * c1),
                       is3 = ok && k == (KT)(3 * c1);
            valid += __popcll(__ballot(ok));
            cc0 += __popcll(__ballot(is0));
            cc1 += __popcll(__ballot(is1));
            cc2 += __popcll(__ballot(is2));
            cc3 += __popcll(__ballot(is3));
            if (ok && !(is0 | is1 | is2 | is3)) {
                u32 h = (u32)(((u64)k * 0x9E3779B97F4A7C15ull) >> 52) & (HIST_SLOTS - 1);
                while (true) {
                    const KT cur = ((volatile KT*)t_key)[h];
                    if (cur == k) break;
                    if (cur == SENT) {
                        const KT old = atomicCAS(&t_key[h], SENT, k);
                        if (old == SENT) {
                            atomicAdd(&n_used, 1u);
                            break;
                        }
                        if (old == k) break;
                    }
                    h = (h + 1) & (HIST_SLOTS - 1);
                }
                atomicAdd(&t_cnt[h], 1u);
            }
        }
        __syncthreads();
        // the table persists across tiles (no per-tile reset); it is flushed to the bins in HBM once it is half full,
        // so that the next tile (at most HIST_TILE new keys) always fits
        if (n_used > HIST_SLOTS / 4) {   // load factor stays <= 0.5 through the next tile (short probe sequences)
            for (int i = threadIdx.x; i < HIST_SLOTS; i += HIST_THREADS) {
                const u32 c = t_cnt[i];
                if (c) atomicAdd(&bins[t_key[i]], c);
                t_key[i] = SENT;
                t_cnt[i] = 0;
            }
            __syncthreads();
            if (threadIdx.x == 0) n_used = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_SLOTS; i += HIST_THREADS) {
        const u32 c = t_cnt[i];
        if (c) atomicAdd(&bins[t_key[i]], c);
    }
    if ((threadIdx.x & 63) == 0) {   // every lane of a wave holds the same wave totals
        if (cc0) atomicAdd(&bins[0], cc0);
        if (cc1) atomicAdd(&bins[(u64)c1], cc1);
        if (cc2) atomicAdd(&bins[(u64)(KT)(2 * c1)], cc2);
        if (cc3) atomicAdd(&bins[(u64)(KT)(3 * c1)], cc3);
        if (valid) atomicAdd(n_valid, valid);   // number of usable sites (fasta.py:57)
    }
}

#define SCAN_BLOCK_ELEMS 4096  // bins per block in the count / write passes

// (also the largest count, into *max_count: it decides the limb count of the int8 Gram and used to cost a D x 4-byte
// copy to the host per alignment)
__global__ __launch_bounds__(256) void k_bins_count(const u32* __restrict__ bins, int64_t nbins,
                                                    u32* __restrict__ block_nz, u32* __restrict__ max_count) {
    __shared__ u32 sh[4], shm[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK_ELEMS;
    u32 c = 0, mx = 0;
    for (int j = 0; j < SCAN_BLOCK_ELEMS / 256 / 4; ++j) {
        const int64_t i = base + ((int64_t)j * 256 + threadIdx.x) * 4;
        if (i + 3 < nbins) {
            const uint4 v = *reinterpret_cast<const uint4*>(bins + i);
            c += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
            mx = max(max(mx, v.x), max(max(v.y, v.z), v.w));
        } else {
            for (int e = 0; e < 4; ++e)
                if (i + e < nbins) {
                    c += bins[i + e] != 0;
                    mx = max(mx, bins[i + e]);
                }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c += __shfl_xor(c, d, 64);
        mx = max(mx, (u32)__shfl_xor((int)mx, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = c;
        shm[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_nz[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        const u32 m = max(max(shm[0], shm[1]), max(shm[2], shm[3]));
        if (m) atomicMax(max_count, m);
    }
}

// exclusive scan of block_nz (one workgroup; nblocks up to a few million) -> offsets, total at [nblocks]
__global__ __launch_bounds__(1024) void k_scan_blocks(const u32* __restrict__ in, int64_t nblocks,
                                                      u64* __restrict__ out) {
    __shared__ u64 sh[16];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = 0; base < nblocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const u64 v = i < nblocks ? in[i] : 0;
        u64 x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) sh[w] = x;
        __syncthreads();
        u64 pre = carry;
        for (int k = 0; k < w; ++k) pre += sh[k];
        if (i < nblocks) out[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[nblocks] = carry;
}

// Ordered compaction of the non-empty bins -> keys ascending, counts, weights = count / N (counts[k] / float(L):
// simulation.py:54, fasta.py:66-70).  `rezero`: every non-empty bin is cleared after it was read (and the two meta words
// behind the bins by block 0), so a pooled bin array goes back all zero.
__global__ __launch_bounds__(256) void k_bins_write(u32* __restrict__ bins, int64_t nbins,
                                                    const u64* __restrict__ block_off, u64* __restrict__ keys,
                                                    u32* __restrict__ counts, double* __restrict__ weights, double N,
                                                    int rezero) {
    __shared__ u32 sh[5];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK_ELEMS;
    // each thread owns 16 consecutive bins -> ordered output
    const int64_t lo = base + (int64_t)threadIdx.x * 16;
    u32 v[16];
    u32 c = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v[e] = (lo + e < nbins) ? bins[lo + e] : 0;
        c += v[e] != 0;
    }
    // block exclusive scan of c
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 x = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    u32 pre = 0;
    for (int k = 0; k < w; ++k) pre += sh[k];
    u64 o = block_off[blockIdx.x] + pre + x - c;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (v[e]) {
            keys[o] = (u64)(lo + e);
            counts[o] = v[e];
            weights[o] = (double)v[e] / N;
            if (rezero) bins[lo + e] = 0;
            ++o;
        }
    if (rezero && blockIdx.x == 0 && threadIdx.x < 4) bins[nbins + threadIdx.x] = 0;   // n_valid, max count
}

__global__ void k_counts_to_weights(const u32* __restrict__ counts, int64_t D, double N, double* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) w[i] = (double)counts[i] / N;  // counts[k] / float(L): simulation.py:54, fasta.py:66-70
}

// ---- sort-based histogram: any number of taxa ---------------------------------------------------------------------------
// The direct bin array costs 3 passes over 4 * 4^n bytes (16 GiB at 16 taxa: 16 ms for a 1 M-site alignment) and does not
// exist beyond 16 taxa.  When the bins outweigh the sites (4^n > 32 L) or n > 16 the site words are radix-sorted on the
// bits in use (csrc/radix_sort.h, hand-written since round 3) and run-length encoded instead: unique keys come out ascending like the bins' compaction, the
// invalid-site marker (all ones: bit 2n set) sorts last and is cut off.
template <typename KT>
__global__ void k_widen_keys(const KT* __restrict__ in, int64_t D, u64* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) out[i] = (u64)in[i];
}

// info[0] = number of runs (k_rle_write); this adds the last run's key (info[2..3]), its length (info[4]) and whether it is
// the invalid-site marker (info[5]).  starts[r] = first position of run r, starts[nr] = number of sites.
template <typename KT>
__global__ void k_last_run(const KT* __restrict__ uniq, const u32* __restrict__ starts, u32* __restrict__ info) {
    const u32 nr = info[0];
    const unsigned long long k = nr ? (unsigned long long)uniq[nr - 1] : 0ull;
    info[2] = (u32)k;
    info[3] = (u32)(k >> 32);
    info[4] = nr ? starts[nr] - starts[nr - 1] : 0u;
    info[5] = (nr && uniq[nr - 1] == (KT)~(KT)0) ? 1u : 0u;
}
// keys widened to 64 bits, counts = lengths of the runs, weights = count / N and the largest count (info[6], zero at launch)
// of the D kept runs
template <typename KT>
__global__ __launch_bounds__(256) void k_table_finish(const KT* __restrict__ uniq, const u32* __restrict__ starts, int64_t D,
                                                      double n_sites, u64* __restrict__ keys, u32* __restrict__ counts,
                                                      double* __restrict__ weights, u32* __restrict__ info) {
    __shared__ u32 wmax[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (i < D) {
        c = starts[i + 1] - starts[i];
        keys[i] = (u64)uniq[i];
        counts[i] = c;
        weights[i] = (double)c / n_sites;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 o = __shfl_xor(c, d, 64);
        c = o > c ? o : c;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per workgroup, and only if its maximum can still matter (one per wave: 30 us)
        const u32 m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m > __hip_atomic_load(&info[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&info[6], m);
    }
}

template <typename KT>
static int build_sorted(sp_ctx* ctx, const KT* dkeys, int64_t L, int n_taxa, sp_alignment** out) {
    SP_REQUIRE(L < ((int64_t)1 << 32), SP_ELIMIT, "sort-based histogram: at most 2^32 - 1 sites per call (got %lld)", (long long)L);
    // work buffers pooled in the context (six hipMalloc / hipFree per alignment cost more than the kernels: 1.7 ms of wall
    // time for 0.37 ms of device time at 16 taxa x 1 M sites)
    DevBuf &sorted = ctx->hist_work[0], &sorted2 = ctx->hist_work[1], &uniq = ctx->hist_work[2], &cnts = ctx->hist_work[3],
           &nruns = ctx->hist_work[4], &tmp = ctx->hist_work[5];
    auto fail = [&](int code) { return code; };
    const size_t l1 = (size_t)std::max<int64_t>(L, 1);
    int rc;
    if ((rc = sorted.ensure(l1 * sizeof(KT))) || (rc = sorted2.ensure(l1 * sizeof(KT))) || (rc = uniq.ensure(l1 * sizeof(KT))) ||
        (rc = cnts.ensure(l1 * 4)) || (rc = nruns.ensure(64)))
        return fail(rc);
    const unsigned end_bit = (unsigned)std::min<int>(2 * n_taxa + 1, 8 * (int)sizeof(KT));   // + 1: the marker's bit
    u32 info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u32* starts = nullptr;   // run starts (in `tmp`, alive until the table is finished)
    if (L > 0) {
        PhaseScope ps(ctx, SP_PHASE_HIST);
        // (radix_sort.h: stable one-sweep LSD passes of 8 bits by wave match + a run-length encode by head flags; rounds
        // 1 - 2 called rocprim::radix_sort_keys / run_length_encode here)
        SP_HIP(hipMemsetAsync(nruns.p, 0, 32, ctx->stream));
        const KT* skeys = nullptr;
        if ((rc = rs_sort<KT>(ctx, dkeys, sorted.as<KT>(), sorted2.as<KT>(), nullptr, nullptr, nullptr, L, 1, end_bit, tmp, &skeys,
                              nullptr)))
            return fail(rc);
        if ((rc = rs_run_length_encode<KT>(ctx, skeys, L, uniq.as<KT>(), nullptr, nruns.as<u32>(), tmp, &starts))) return fail(rc);
        hipLaunchKernelGGL(k_last_run<KT>, dim3(1), dim3(1), 0, ctx->stream, uniq.as<KT>(), starts, nruns.as<u32>());
        hipError_t e = hipMemcpyAsync(info, nruns.p, 32, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            sp_set_error("sort-based histogram failed: %s", hipGetErrorString(e));
            return fail(SP_EHIP);
        }
    }
    const u32 nr = info[0];
    int64_t D = nr, dropped = 0;
    // the last run is the invalid-site marker if there were invalid sites
    if (nr > 0 && info[5] && 2 * n_taxa < 8 * (int)sizeof(KT)) {
        D -= 1;
        dropped = info[4];
    }
    sp_alignment* al = new sp_alignment();
    al->ctx = ctx;
    al->n_taxa = n_taxa;
    al->D = D;
    al->N = L - dropped;
    al->exact = true;
    const size_t d1 = (size_t)std::max<int64_t>(D, 1);
    if ((rc = al->keys.ensure(d1 * 8)) || (rc = al->weights.ensure(d1 * 8)) || (rc = al->counts.ensure(d1 * 4 + SP_COUNTS_PAD))) {
        sp_alignment_destroy(al);
        return fail(rc);
    }
    hipError_t e = hipSuccess;
    if (D > 0) {
        PhaseScope ps(ctx, SP_PHASE_HIST);
        hipLaunchKernelGGL(k_table_finish<KT>, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, uniq.as<KT>(),
                           starts, D, (double)al->N, al->keys.as<u64>(), al->counts.as<u32>(), al->weights.as<double>(),
                           nruns.as<u32>());
        e = hipMemcpyAsync(info, nruns.p, 32, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipGetLastError();
        al->max_count = info[6];   // largest count -> limb count of the int8 Gram
    }
    if (e != hipSuccess) {
        sp_alignment_destroy(al);
        sp_set_error("sort-based histogram (compaction): %s", hipGetErrorString(e));
        return SP_EHIP;
    }
    *out = al;
    return SP_OK;
}

static int build_from_device_keys(sp_ctx* ctx, const void* dkeys, bool keys32, int64_t L, int n_taxa,
                                  sp_alignment** out) {
    {
        const int force = ctx->opt.hist_sort;   // 1 / 0: force / forbid the sort-based form (test switch), -1: auto
        const bool big_bins = n_taxa > 16 || pow4(n_taxa) > 32 * std::max<int64_t>(L, 1);
        const bool use_sort = n_taxa > 16 || (force >= 0 ? force == 1 : big_bins);
        if (use_sort)
            return keys32 ? build_sorted<u32>(ctx, (const u32*)dkeys, L, n_taxa, out)
                          : build_sorted<u64>(ctx, (const u64*)dkeys, L, n_taxa, out);
    }
    const int64_t nbins = pow4(n_taxa);
    const int64_t nblocks = (nbins + SCAN_BLOCK_ELEMS - 1) / SCAN_BLOCK_ELEMS;
    // Up to 12 taxa (64 MB of bins) the three work arrays belong to the context and the bins are handed back all zero by
    // k_bins_write: no allocation and no memset per alignment (they were most of the call's wall time: 0.42 ms around
    // 0.04 ms of kernels at config 2).  Larger bin arrays are allocated, cleared and freed per call as before.
    const bool pooled = nbins * 4 <= ((int64_t)64 << 20);
    DevBuf l_bins, l_blk, l_off;
    DevBuf& bins = pooled ? ctx->hist_bins : l_bins;
    DevBuf& blk = pooled ? ctx->hist_blk : l_blk;
    DevBuf& off = pooled ? ctx->hist_off : l_off;
    int rc;
    auto cleanup = [&]() {
        l_bins.release();
        l_blk.release();
        l_off.release();
    };
    const void* had = bins.p;
    const size_t had_cap = bins.cap;
    if ((rc = bins.ensure((size_t)nbins * 4 + 16)) || (rc = blk.ensure((size_t)nblocks * 4)) ||
        (rc = off.ensure((size_t)(nblocks + 1) * 8))) {
        cleanup();
        return rc;
    }
    if (pooled && (bins.p != had || bins.cap != had_cap)) ctx->hist_clean = false;   // a fresh (larger) array
    u32* n_valid = bins.as<u32>() + nbins;      // the words after the bins: usable sites, largest count
    u32* max_dev = n_valid + 1;
    {
        PhaseScope ps(ctx, SP_PHASE_HIST);
        if (!(pooled && ctx->hist_clean) &&
            hipMemsetAsync(bins.p, 0, (size_t)nbins * 4 + 16, ctx->stream) != hipSuccess) {
            cleanup();
            sp_set_error("hipMemsetAsync of the bin array failed");
            return SP_EHIP;
        }
        ctx->hist_clean = false;   // dirty from here until k_bins_write has run to completion
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((L + HIST_TILE - 1) / HIST_TILE, ctx->n_cu * 2));
        if (keys32)   // 32-bit site words (n <= 15): half the traffic of the pass (4 L bytes)
            hipLaunchKernelGGL(k_hist_lds<u32>, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, (const u32*)dkeys, L,
                               n_taxa, bins.as<u32>(), n_valid);
        else
            hipLaunchKernelGGL(k_hist_lds<u64>, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, (const u64*)dkeys, L,
                               n_taxa, bins.as<u32>(), n_valid);
        hipLaunchKernelGGL(k_bins_count, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, bins.as<u32>(), nbins,
                           blk.as<u32>(), max_dev);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, blk.as<u32>(), nblocks, off.as<u64>());
    }
    u64 D64 = 0;
    u32 meta[2] = {0, 0};   // usable sites, largest count
    hipError_t e = hipMemcpyAsync(&D64, off.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(meta, n_valid, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        cleanup();
        sp_set_error("histogram failed: %s", hipGetErrorString(e));
        return SP_EHIP;
    }
    const int64_t D = (int64_t)D64;
    const u32 N32 = meta[0];
    sp_alignment* al = new sp_alignment();
    al->ctx = ctx;
    al->n_taxa = n_taxa;
    al->D = D;
    al->N = N32;
    al->exact = true;
    al->max_count = meta[1];
    const size_t d1 = (size_t)std::max<int64_t>(D, 1);
    if ((rc = al->keys.ensure(d1 * 8)) || (rc = al->weights.ensure(d1 * 8)) || (rc = al->counts.ensure(d1 * 4 + SP_COUNTS_PAD))) {
        cleanup();
        sp_alignment_destroy(al);
        return rc;
    }
    if (D > 0) {
        PhaseScope ps(ctx, SP_PHASE_HIST);
        hipLaunchKernelGGL(k_bins_write, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, bins.as<u32>(), nbins,
                           off.as<u64>(), al->keys.as<u64>(), al->counts.as<u32>(), al->weights.as<double>(), (double)N32,
                           pooled ? 1 : 0);
    } else if (pooled) {
        // no usable site: nothing was added to the bins; the meta words are cleared by the next call's memset
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    cleanup();
    if (e != hipSuccess) {
        sp_alignment_destroy(al);
        sp_set_error("histogram compaction failed: %s", hipGetErrorString(e));
        return SP_EHIP;
    }
    if (pooled && D > 0) ctx->hist_clean = true;   // every bin that was touched has been cleared again
    *out = al;
    return SP_OK;
}

extern "C" int sp_alignment_from_site_keys(sp_ctx* ctx, const uint64_t* site_keys, int64_t L, int n_taxa,
                                           sp_alignment** out) {
    return sp_guard("sp_alignment_from_site_keys", [&]() -> int {
    SP_REQUIRE(ctx && out && (site_keys || L == 0), SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 32 && L >= 0, SP_EINVAL, "bad n_taxa / L");
    SP_HIP(hipSetDevice(ctx->device));
    const bool keys32 = n_taxa <= 15;
    SP_CHECK(ctx->misc.ensure((size_t)std::max<int64_t>(L, 1) * (keys32 ? 4 : 8)));
    if (L > 0) {
        if (keys32) {   // narrow on the host: halves the PCIe transfer and the HBM traffic of the histogram pass
            std::vector<u32> k32((size_t)L);
            const u64 lim = (1ull << (2 * n_taxa)) - 1;
            for (int64_t i = 0; i < L; ++i) {
                SP_REQUIRE(site_keys[i] <= lim || site_keys[i] == ~0ull, SP_EINVAL,
                           "site key %llu does not fit %d taxa", (unsigned long long)site_keys[i], n_taxa);
                k32[i] = site_keys[i] == ~0ull ? 0xFFFFFFFFu : (u32)site_keys[i];
            }
            SP_HIP(hipMemcpyAsync(ctx->misc.p, k32.data(), (size_t)L * 4, hipMemcpyHostToDevice, ctx->stream));
            SP_HIP(hipStreamSynchronize(ctx->stream));   // k32 is a host temporary
        } else {
            SP_HIP(hipMemcpyAsync(ctx->misc.p, site_keys, (size_t)L * 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    return build_from_device_keys(ctx, ctx->misc.p, keys32, L, n_taxa, out);
    });
}

extern "C" int sp_alignment_from_sequences(sp_ctx* ctx, const uint8_t* seqs, int n_taxa, int64_t L, int64_t stride,
                                           sp_alignment** out) {
    return sp_guard("sp_alignment_from_sequences", [&]() -> int {
    SP_REQUIRE(ctx && out && (seqs || L == 0), SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 32 && L >= 0 && stride >= L, SP_EINVAL, "bad n_taxa / L / stride");
    SP_HIP(hipSetDevice(ctx->device));
    SP_CHECK(ctx->misc.ensure((size_t)std::max<int64_t>(L, 1) * 8));
    SP_CHECK(ctx->misc2.ensure((size_t)std::max<int64_t>(L, 1) * n_taxa));
    if (L > 0) {
        SP_HIP(hipMemcpy2DAsync(ctx->misc2.p, (size_t)L, seqs, (size_t)stride, (size_t)L, (size_t)n_taxa,
                                hipMemcpyHostToDevice, ctx->stream));
        PhaseScope ps(ctx, SP_PHASE_HIST);
        if (n_taxa <= 15)
            hipLaunchKernelGGL(k_pack_sites<u32>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->misc2.as<uint8_t>(), n_taxa, L, L, ctx->misc.as<u32>());
        else
            hipLaunchKernelGGL(k_pack_sites<u64>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->misc2.as<uint8_t>(), n_taxa, L, L, ctx->misc.as<u64>());
        SP_HIP(hipGetLastError());
    }
    return build_from_device_keys(ctx, ctx->misc.p, n_taxa <= 15, L, n_taxa, out);
    });
}

// ------------------------------------------------------------------------------------------------------------------
// Alignment simulator: replaces splitp/simulation.py:9-56 (evolve_pattern / generate_alignment) - an
// independent Markov walk down the tree per site: uniform root state (simulation.py:28), at every node the new state
// is drawn from column `state` of that node's 4 x 4 transition matrix (simulation.py:17-18, probs = M[:, index(state)],
// random.choices = inverse CDF over the cumulative weights), leaves write their state into the site's pattern.
// One thread per site, the node states of the walk packed 2 bits each in registers, the cumulative columns of all
// transition matrices in LDS, a counter-based generator (SplitMix64 of seed, site and node - any site can be
// regenerated on its own, results do not depend on the launch shape).  The site words go straight into the histogram
// above: L sites -> D (pattern, count) pairs without leaving the device.
#define SIM_MAX_NODES 64

__device__ __forceinline__ u64 sim_mix(u64 x) {   // SplitMix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <typename KT>
__global__ __launch_bounds__(256) void k_simulate_sites(int n_nodes, const int* __restrict__ parent,
                                                         const int* __restrict__ leaf_taxon,
                                                         const double* __restrict__ trans, int n_taxa, int64_t L,
                                                         u64 seed, KT* __restrict__ site_keys) {
    __shared__ double cum[SIM_MAX_NODES][4][3];   // cumulative probabilities of the first three new states, per old state
    __shared__ int par[SIM_MAX_NODES], leaf[SIM_MAX_NODES];
    for (int i = threadIdx.x; i < n_nodes * 4; i += blockDim.x) {
        const int node = i >> 2, old = i & 3;
        const double* m = trans + (size_t)node * 16;   // row-major M[new][old]
        const double p0 = m[0 * 4 + old], p1 = m[1 * 4 + old], p2 = m[2 * 4 + old];
        cum[node][old][0] = p0;
        cum[node][old][1] = p0 + p1;
        cum[node][old][2] = (p0 + p1) + p2;
    }
    for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) {
        par[i] = parent[i];
        leaf[i] = leaf_taxon[i];
    }
    __syncthreads();
    for (int64_t site = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; site < L; site += (int64_t)gridDim.x * blockDim.x) {
        const u64 base = sim_mix(seed ^ sim_mix((u64)site));
        u64 st_lo = 0, st_hi = 0;   // 2 bits per node
        KT key = 0;
        for (int node = 0; node < n_nodes; ++node) {
            const u64 rnd = sim_mix(base + (u64)node * 0xD1B54A32D192ED03ull);
            const double u = (double)(rnd >> 11) * (1.0 / 9007199254740992.0);   // [0, 1)
            int s;
            const int p = par[node];
            if (p < 0) {
                s = (int)(rnd >> 62);                  // root: uniform over the four states
            } else {
                const int old = (int)(((p < 32 ? st_lo >> (2 * p) : st_hi >> (2 * (p - 32)))) & 3ull);
                const double* c = cum[node][old];
                s = (u >= c[0]) + (u >= c[1]) + (u >= c[2]);
            }
            if (node < 32)
                st_lo |= (u64)s << (2 * node);
            else
                st_hi |= (u64)s << (2 * (node - 32));
            const int t = leaf[node];
            if (t >= 0) key |= (KT)s << (2 * (n_taxa - 1 - t));
        }
        site_keys[site] = key;
    }
}

extern "C" int sp_simulate_alignment(sp_ctx* ctx, int n_nodes, const int32_t* parent, const int32_t* leaf_taxon,
                                     const double* transition, int n_taxa, int64_t L, uint64_t seed,
                                     sp_alignment** out) {
    return sp_guard("sp_simulate_alignment", [&]() -> int {
    SP_REQUIRE(ctx && parent && leaf_taxon && transition && out, SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 31, SP_ELIMIT, "device simulator supports 2..31 taxa (got %d)", n_taxa);
    SP_REQUIRE(n_nodes >= n_taxa && n_nodes <= SIM_MAX_NODES, SP_ELIMIT, "tree has %d nodes (supported: n_taxa..%d)",
               n_nodes, SIM_MAX_NODES);
    SP_REQUIRE(L >= 0, SP_EINVAL, "L < 0");
    std::vector<int> seen((size_t)n_taxa, 0);
    int roots = 0;
    for (int i = 0; i < n_nodes; ++i) {
        SP_REQUIRE(parent[i] < i, SP_EINVAL, "nodes must be listed parents first (node %d has parent %d)", i, parent[i]);
        roots += parent[i] < 0;
        SP_REQUIRE(leaf_taxon[i] < n_taxa, SP_EINVAL, "leaf_taxon[%d] = %d out of range", i, leaf_taxon[i]);
        if (leaf_taxon[i] >= 0) seen[leaf_taxon[i]] += 1;
        for (int k = 0; k < 16 && parent[i] >= 0; ++k)
            SP_REQUIRE(transition[(size_t)i * 16 + k] >= 0.0, SP_EINVAL, "negative transition probability at node %d", i);
    }
    SP_REQUIRE(roots == 1 && parent[0] < 0, SP_EINVAL, "exactly one root, listed first");
    for (int t = 0; t < n_taxa; ++t) SP_REQUIRE(seen[t] == 1, SP_EINVAL, "taxon %d is the leaf of %d nodes", t, seen[t]);
    SP_HIP(hipSetDevice(ctx->device));
    const bool keys32 = n_taxa <= 15;
    SP_CHECK(ctx->misc.ensure((size_t)std::max<int64_t>(L, 1) * (keys32 ? 4 : 8)));
    SP_CHECK(ctx->misc2.ensure((size_t)n_nodes * (16 * 8 + 8)));
    double* d_trans = ctx->misc2.as<double>();
    int* d_parent = reinterpret_cast<int*>(d_trans + (size_t)n_nodes * 16);
    int* d_leaf = d_parent + n_nodes;
    SP_HIP(hipMemcpyAsync(d_trans, transition, (size_t)n_nodes * 16 * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(d_parent, parent, (size_t)n_nodes * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(d_leaf, leaf_taxon, (size_t)n_nodes * 4, hipMemcpyHostToDevice, ctx->stream));
    if (L > 0) {
        PhaseScope ps(ctx, SP_PHASE_HIST);
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((L + 255) / 256, (int64_t)ctx->n_cu * 16));
        if (keys32)
            hipLaunchKernelGGL(k_simulate_sites<u32>, dim3(grid), dim3(256), 0, ctx->stream, n_nodes, d_parent, d_leaf,
                               d_trans, n_taxa, L, (u64)seed, ctx->misc.as<u32>());
        else
            hipLaunchKernelGGL(k_simulate_sites<u64>, dim3(grid), dim3(256), 0, ctx->stream, n_nodes, d_parent, d_leaf,
                               d_trans, n_taxa, L, (u64)seed, ctx->misc.as<u64>());
        SP_HIP(hipGetLastError());
    }
    SP_HIP(hipStreamSynchronize(ctx->stream));   // the host arrays may die at return
    return build_from_device_keys(ctx, ctx->misc.p, keys32, L, n_taxa, out);
    });
}

// Test entry (tests/test_gpu_direct.py::test_radix_sort_direct): the library's stable segmented radix sort on host
// arrays - keys as 64-bit words (narrowed to 32 bits when key_bytes == 4), optional 32-bit values carried along - so that
// segment counts, ragged tile ends and key widths the histogram / big-table callers never produce are checked against a
// reference sort directly.  n = seg_len * n_seg elements in and out.
extern "C" int sp_debug_radix_sort(sp_ctx* ctx, const uint64_t* keys_host, const uint32_t* vals_host, int key_bytes,
                                   int64_t seg_len, int64_t n_seg, unsigned end_bit, uint64_t* keys_out, uint32_t* vals_out) {
    return sp_guard("sp_debug_radix_sort", [&]() -> int {
    SP_REQUIRE(ctx && keys_host && keys_out && (key_bytes == 4 || key_bytes == 8) && seg_len >= 0 && n_seg >= 0, SP_EINVAL,
               "sp_debug_radix_sort: bad argument");
    SP_REQUIRE(!vals_host == !vals_out, SP_EINVAL, "sp_debug_radix_sort: values in and out go together");
    SP_HIP(hipSetDevice(ctx->device));
    const int64_t n = seg_len * n_seg;
    if (n == 0) return SP_OK;
    DevBuf kin, ka, kb, vin, va, vb, work;
    auto bail = [&](int code) {
        kin.release(); ka.release(); kb.release(); vin.release(); va.release(); vb.release(); work.release();
        return code;
    };
    int rc;
    if ((rc = kin.ensure((size_t)n * 8)) || (rc = ka.ensure((size_t)n * 8)) || (rc = kb.ensure((size_t)n * 8)) ||
        (rc = vin.ensure((size_t)n * 4)) || (rc = va.ensure((size_t)n * 4)) || (rc = vb.ensure((size_t)n * 4)))
        return bail(rc);
    if (key_bytes == 4) {
        std::vector<u32> k32((size_t)n);
        for (int64_t i = 0; i < n; ++i) k32[i] = (u32)keys_host[i];
        if (hipMemcpy(kin.p, k32.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(SP_EHIP);
    } else if (hipMemcpy(kin.p, keys_host, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) {
        return bail(SP_EHIP);
    }
    if (vals_host && hipMemcpy(vin.p, vals_host, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(SP_EHIP);
    const u32* sv = nullptr;
    if (key_bytes == 4) {
        const u32* sk = nullptr;
        rc = rs_sort<u32>(ctx, kin.as<u32>(), ka.as<u32>(), kb.as<u32>(), vals_host ? vin.as<u32>() : nullptr, va.as<u32>(),
                          vb.as<u32>(), seg_len, n_seg, end_bit, work, &sk, vals_host ? &sv : nullptr);
        if (rc != SP_OK) return bail(rc);
        std::vector<u32> k32((size_t)n);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(k32.data(), sk, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(SP_EHIP);
        for (int64_t i = 0; i < n; ++i) keys_out[i] = k32[i];
    } else {
        const u64* sk = nullptr;
        rc = rs_sort<u64>(ctx, kin.as<u64>(), ka.as<u64>(), kb.as<u64>(), vals_host ? vin.as<u32>() : nullptr, va.as<u32>(),
                          vb.as<u32>(), seg_len, n_seg, end_bit, work, &sk, vals_host ? &sv : nullptr);
        if (rc != SP_OK) return bail(rc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(keys_out, sk, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(SP_EHIP);
    }
    if (vals_host && hipMemcpy(vals_out, sv, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return bail(SP_EHIP);
    return bail(SP_OK);
    });
}
