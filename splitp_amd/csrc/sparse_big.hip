// Big-table form of the sparse route (k_sparse_big): count tables beyond the 65535 rows of the list kernels and
// float-weight tables at 12+ taxa - everything in global memory, list orders from segmented radix sorts.
// Same block iteration, stop rule and Cholesky-QR as sparse.hip (sparse_common.h).
#include "sparse_common.h"

#include "radix_sort.h"   // stable segmented radix sort: both list orders of every split in one call each

// =====================================================================================================================
// Big-table form: count tables with more than 65535 patterns (the 16-bit ids / offsets of the list kernels above end
// there; 12 taxa beyond ~1 M sites, long branches), any side size, everything in global memory.
// Per split the dense route's reindex kernel has already produced compact (row, col) for every pattern.  The two list
// orders come from ONE stable segmented radix sort each (csrc/radix_sort.h; segments = splits; keys = col resp. row, values =
// pattern index), so every group is in table order like the counting sort above: reproducible bit for bit.
// Products walk a sorted order in 1024 thread-chunks: a thread sums the runs inside its chunk; a run that crosses chunk
// borders leaves partial sums (first / last run of a chunk) in LDS and its head's owner adds them up in chunk order.
// Work is balanced whatever the group sizes (a 2|10 split has 16 rows of 6 k entries each).  Same block iteration, stop
// rule and Cholesky-QR as above (V, W column-major in a per-workgroup slab); persistent workgroups loop over splits.
#define SPKB_MAXHALF 40

struct SpkbPart {
    double first[SPK_THREADS][4];   // partial sums of a chunk's first run when it continues a previous chunk's run
    double last[SPK_THREADS][4];    // ... of its last run when that continues into the next chunk
    unsigned int flags[SPK_THREADS];   // bit 0: chunk starts inside a run; bit 1: the WHOLE chunk is inside that run
};

// out[key][0..3] = sum over the entries j of the run `key` of count * in[minor][0..3]; keys sorted (stable), perm = pattern
// index of every sorted position, minor_of / counts indexed by pattern.  out has `nmajor` rows.  Ends with a barrier.
// RM: `in` / `out` are row-major blocks ([row][4]: the four values of a row are ONE 32-byte access - a column-major block
// costs four L2 sectors per entry, and the products of a 124 k-pattern table are L2-bandwidth bound); else column-major
// with column strides ics / ocs (the 8-wide fallback block, two 4-column passes).
template <typename CT, bool RM>
__device__ __forceinline__ void spkb_product(const u32* __restrict__ keys, const u32* __restrict__ minor,
                                             const CT* __restrict__ cnt, int D, const double* __restrict__ in, int ics,
                                             double* __restrict__ out, int ocs, int nmajor, SpkbPart& pt) {
    auto iat = [&](u32 row, int k) { return RM ? (size_t)row * 4 + k : (size_t)k * ics + row; };
    auto oat = [&](u32 row, int k) { return RM ? (size_t)row * 4 + k : (size_t)k * ocs + row; };
    for (int i = threadIdx.x; i < nmajor; i += SPK_THREADS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[oat((u32)i, k)] = 0.0;
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int chunk = (D + SPK_THREADS - 1) / SPK_THREADS;
    const int lo = min(D, t * chunk), hi = min(D, lo + chunk);
    unsigned int fl = 0;
    // keys / minor / cnt hold the sorted order CHUNK-INTERLEAVED (k_gather_sorted): element jj of thread t's chunk sits at
    // jj * SPK_THREADS + t, so the 64 lanes of a load read 64 consecutive words although every thread walks its own
    // contiguous piece of the sorted order (thread-contiguous addressing measured 120 ms for 2035 splits of a 124 k table).
    auto at = [&](int thread, int jj) { return (size_t)jj * SPK_THREADS + thread; };
    if (lo < hi) {
        double acc[4] = {0, 0, 0, 0};
        u32 cur = keys[at(t, 0)];
        bool first_run = lo > 0 && keys[at(t - 1, chunk - 1)] == cur;   // continuation of an earlier chunk's run
        if (first_run) fl = 1;
#pragma unroll 4
        for (int jj = 0; jj < hi - lo; ++jj) {
            const u32 key = keys[at(t, jj)];
            const u32 m = minor[at(t, jj)];
            const double c = (double)cnt[at(t, jj)];
            double x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = in[iat(m, k)];
            if (key != cur) {   // the run `cur` ended inside the chunk
                if (first_run) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) pt.first[t][k] = acc[k];
                    first_run = false;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) out[oat(cur, k)] = acc[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = 0.0;
                cur = key;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fma(c, x[k], acc[k]);
        }
        const bool goes_on = hi < D && keys[at(t + 1, 0)] == cur;
        if (first_run) {           // the whole chunk was inside the run it started in
#pragma unroll
            for (int k = 0; k < 4; ++k) pt.first[t][k] = acc[k];
            if (goes_on) fl |= 2;
        } else if (goes_on) {      // head in this chunk, tail in the next: finished by the fix-up below
#pragma unroll
            for (int k = 0; k < 4; ++k) pt.last[t][k] = acc[k];
            fl |= 4;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[oat(cur, k)] = acc[k];
        }
    }
    pt.flags[t] = fl;
    __syncthreads();
    if (fl & 4) {   // this thread owns the head of a run that crosses chunk borders
        const u32 key = keys[at(t, hi - lo - 1)];
        double acc[4] = {pt.last[t][0], pt.last[t][1], pt.last[t][2], pt.last[t][3]};
        for (int u = t + 1; u < SPK_THREADS; ++u) {
            const unsigned int fu = pt.flags[u];
            if (!(fu & 1)) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += pt.first[u][k];
            if (!(fu & 2)) break;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) out[oat(key, k)] = acc[k];
    }
    __syncthreads();
}

// Sorted order of every split -> chunk-interleaved streams for spkb_product: position j of a segment (thread j / chunk,
// element j % chunk) goes to (j % chunk) * SPK_THREADS + j / chunk of the segment's padded block of chunk * SPK_THREADS.
// That is the transpose of a [SPK_THREADS][chunk] matrix, done in 32 x 32 tiles through LDS (round 4): a workgroup reads 32
// runs of 32 consecutive sorted positions (coalesced: key, pattern index), fetches the pattern's other coordinate and count
// (the two gathers nothing avoids) and writes 32 runs of 32 consecutive destination words.  (Round 3's form - one thread per
// destination word, every one of its four reads 122 words from its neighbour's - took 1.5 ms per 84 M entries, 9 of the
// 48 ms of a 2035-split call on a 124 k-pattern table.)
#define GS_T 32
template <typename CT>
__global__ __launch_bounds__(256) void k_gather_sorted(const u32* __restrict__ key_sorted, const u32* __restrict__ perm,
                                                       const u32* __restrict__ other_by_pattern, const CT* __restrict__ counts,
                                                       int64_t D, int n_ct, int64_t S, u32* __restrict__ key_i,
                                                       u32* __restrict__ minor_i, CT* __restrict__ cnt_i) {
    __shared__ u32 tk[GS_T][GS_T + 1], tm[GS_T][GS_T + 1];
    __shared__ CT tc[GS_T][GS_T + 1];
    const int64_t chunk = (D + SPK_THREADS - 1) / SPK_THREADS;
    const int64_t dpad = chunk * SPK_THREADS;
    // grid: x = tile of the chunk axis (n_ct = ceil(chunk / 32)) + 32-thread tile (SPK_THREADS / 32) * n_ct, y = segment
    const int ct = (int)(blockIdx.x % (unsigned)n_ct), tt = (int)(blockIdx.x / (unsigned)n_ct);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
    for (int64_t seg = blockIdx.y; seg < S; seg += gridDim.y) {
#pragma unroll
    for (int r = 0; r < GS_T; r += 8) {
        const int64_t thr = (int64_t)tt * GS_T + ly + r, c = (int64_t)ct * GS_T + lx;   // sorted position j = thr * chunk + c
        const int64_t j = thr * chunk + c;
        u32 k = 0, m = 0;
        CT cv = (CT)0;
        if (c < chunk && j < D) {
            const int64_t g = seg * D + j;
            const u32 pidx = perm[g];
            k = key_sorted[g];
            m = other_by_pattern[seg * D + pidx];
            cv = counts[pidx];
        }
        tk[ly + r][lx] = k;
        tm[ly + r][lx] = m;
        tc[ly + r][lx] = cv;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GS_T; r += 8) {
        const int64_t c = (int64_t)ct * GS_T + ly + r, thr = (int64_t)tt * GS_T + lx;   // destination (c, thr): c * SPK_THREADS + thr
        if (c < chunk && thr * chunk + c < D) {
            const int64_t dst = seg * dpad + c * SPK_THREADS + thr;
            key_i[dst] = tk[lx][ly + r];
            minor_i[dst] = tm[lx][ly + r];
            cnt_i[dst] = tc[lx][ly + r];
        }
    }
    __syncthreads();   // (the tiles are reused by the next segment of this workgroup)
    }
}

// CT = u32: count table (trace exact in u64);  CT = double: float-weight table (trace summed in a fixed tree).
// WIDE = false: the 4-wide certified iteration; a split without a certified gap after SPKB_MAXHALF half products leaves
// with status bit 1.  WIDE = true: the second launch - the 8-wide fallback block for exactly those splits (a workgroup whose
// splits all certified returns at once).  Two kernels instead of one body (round 4): with the wide block's Jacobi and 8-column
// passes inlined next to the hot loop the kernel spilled 585 vector registers (1.1 KB of scratch per lane); apart, the
// 4-wide kernel spills none in its loops.
template <typename CT, bool WIDE>
__global__ __launch_bounds__(SPK_THREADS) void k_sparse_big(int64_t D64, int S, const u32* __restrict__ rr_all,
                                                            const u32* __restrict__ keyc_all, const u32* __restrict__ minc_all,
                                                            const CT* __restrict__ cntc_all,
                                                            const u32* __restrict__ keyr_all, const u32* __restrict__ minr_all,
                                                            const CT* __restrict__ cntr_all,
                                                            const CT* __restrict__ counts, const int2* __restrict__ dims,
                                                            double* __restrict__ slabs, size_t slab_doubles,
                                                            double* __restrict__ scores, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    SpkShared& sh = *reinterpret_cast<SpkShared*>(smem_b);
    SpkbPart& pt = *reinterpret_cast<SpkbPart*>(smem_b + ((sizeof(SpkShared) + 15) & ~(size_t)15));
    EigShared& esh = *reinterpret_cast<EigShared*>(smem_b + ((sizeof(SpkShared) + 15) & ~(size_t)15) +
                                                   ((sizeof(SpkbPart) + 15) & ~(size_t)15));
    const size_t lds_used_b = ((sizeof(SpkShared) + 15) & ~(size_t)15) + ((sizeof(SpkbPart) + 15) & ~(size_t)15) +
                              ((sizeof(EigShared) + 15) & ~(size_t)15) + 16;
    const int D = (int)D64;
    double* slab = slabs + (size_t)blockIdx.x * slab_doubles;
    if (WIDE) {   // anything to do?  (uniform: every thread reads the same status words)
        bool any = false;
        for (int sid = blockIdx.x; sid < S; sid += gridDim.x) any = any || (status[sid] & 2) != 0;
        if (!any) return;
    }
    // trace = sum count^2 (exact in u64 for counts) and the 4 heaviest patterns, once per workgroup
    double trace;
    if (std::is_same<CT, u32>::value) {
        unsigned long long tr_part = 0;
        for (int i = threadIdx.x; i < D; i += SPK_THREADS) tr_part += (unsigned long long)counts[i] * (unsigned long long)counts[i];
        unsigned long long* red64 = reinterpret_cast<unsigned long long*>(pt.first);
        red64[threadIdx.x] = tr_part;
        __syncthreads();
        for (int sft = SPK_THREADS / 2; sft >= 1; sft >>= 1) {
            if ((int)threadIdx.x < sft) red64[threadIdx.x] += red64[threadIdx.x + sft];
            __syncthreads();
        }
        trace = (double)red64[0];
    } else {
        double tr_part = 0;
        for (int i = threadIdx.x; i < D; i += SPK_THREADS) tr_part += (double)counts[i] * (double)counts[i];
        double* redd = reinterpret_cast<double*>(pt.first);
        redd[threadIdx.x] = tr_part;
        __syncthreads();
        for (int sft = SPK_THREADS / 2; sft >= 1; sft >>= 1) {
            if ((int)threadIdx.x < sft) redd[threadIdx.x] += redd[threadIdx.x + sft];
            __syncthreads();
        }
        trace = redd[0];
    }
    __syncthreads();
    int top_idx[4];
    {   // four rounds of a block arg-max over (weight, lowest index)
        double* bval = reinterpret_cast<double*>(pt.first);
        int* bidx = reinterpret_cast<int*>(pt.last);
        int taken[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            double mv = -1.0;
            int mi = -1;
            for (int i = threadIdx.x; i < D; i += SPK_THREADS) {
                const double v = (double)counts[i];
                if (i != taken[0] && i != taken[1] && i != taken[2] && i != taken[3] && v > mv) {   // (ascending i: ties keep the lowest)
                    mv = v;
                    mi = i;
                }
            }
            bval[threadIdx.x] = mv;
            bidx[threadIdx.x] = mi;
            __syncthreads();
            for (int sft = SPK_THREADS / 2; sft >= 1; sft >>= 1) {
                if ((int)threadIdx.x < sft) {
                    const double ov = bval[threadIdx.x + sft];
                    const int oi = bidx[threadIdx.x + sft];
                    const double cv = bval[threadIdx.x];
                    const int ci = bidx[threadIdx.x];
                    if (ov > cv || (ov == cv && oi >= 0 && (ci < 0 || oi < ci))) {
                        bval[threadIdx.x] = ov;
                        bidx[threadIdx.x] = oi;
                    }
                }
                __syncthreads();
            }
            taken[round] = bidx[0];
            top_idx[round] = (bval[0] > 0) ? bidx[0] : -1;
            __syncthreads();
        }
    }
    for (int sid = blockIdx.x; sid < S; sid += gridDim.x) {
        const int R = dims[sid].x, C = dims[sid].y;
        int it_before = 0;
        if (WIDE) {
            const int st_in = status[sid];
            if (!(st_in & 2)) continue;
            it_before = st_in >> 8;
        }
        if (min(R, C) <= 4 || !(trace > 0)) {
            if (threadIdx.x == 0) {
                scores[sid] = trace > 0 ? 0.0 : __builtin_nan("");
                status[sid] = 0;
            }
            continue;
        }
        const u32* rr = rr_all + (size_t)sid * D;
        const size_t dpad = (size_t)((D + SPK_THREADS - 1) / SPK_THREADS) * SPK_THREADS;   // chunk-interleaved blocks
        const u32* keyc = keyc_all + (size_t)sid * dpad;
        const u32* minc = minc_all + (size_t)sid * dpad;
        const CT* cntc = cntc_all + (size_t)sid * dpad;
        const u32* keyr = keyr_all + (size_t)sid * dpad;
        const u32* minr = minr_all + (size_t)sid * dpad;
        const CT* cntr = cntr_all + (size_t)sid * dpad;
        const int Vp = ((R + 3) & ~3) + 4, Wp = ((C + 3) & ~3) + 4;
        // slab: row-major 4-wide V and W ([row][4]) for the normal iteration, then the column-major 8-column blocks of
        // the wide fallback
        const size_t dpad16 = (size_t)D + 16;
        // a block of up to ~2100 rows lives in the free LDS instead (V first, W if it still fits): the product that
        // gathers from it then stays off the L2, which is what bounds this kernel (most splits have one short side)
        double* const lds_blk = reinterpret_cast<double*>(smem_b + lds_used_b);
        const size_t lds_free_d = ((size_t)SPK_LDS_BYTES - lds_used_b) / 8;
        const size_t v_need = (size_t)4 * ((size_t)R + 4), w_need = (size_t)4 * ((size_t)C + 4);
        const bool v_lds = v_need <= lds_free_d;
        const bool w_lds = w_need <= lds_free_d - (v_lds ? v_need : 0);
        double* V = v_lds ? lds_blk : slab;
        double* W = w_lds ? lds_blk + (v_lds ? v_need : 0) : slab + 4 * dpad16;
        double* V8 = slab + 8 * dpad16;
        double* W8 = V8 + (size_t)SPK_WB * Vp;
        // start block: unit vectors on the rows of the 4 most frequent patterns + hash noise, orthonormalised
        int top_row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) top_row[k] = top_idx[k] >= 0 ? (int)rr[top_idx[k]] : -1;
        for (int i = threadIdx.x; i < Vp; i += SPK_THREADS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double x = 0.0;
                if (i < R) {
                    x = 0.02 * spk_hash((unsigned)i, (unsigned)k);
                    bool hit = top_row[k] == i;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hit = hit && !(q < k && top_row[q] == top_row[k]);   // a row only once
                    if (hit) x += 1.0;
                }
                if (i < R) V[(size_t)i * 4 + k] = x;
            }
        }
        __syncthreads();
        spk_gram(V, R, 4, 1, sh);
        spk_chol_factor(sh, false);
        spk_orth(V, R, 4, 1, sh);
        double prev_sum = 0, prev_delta = 0, prev_ratio = 1.0, top4 = 0, prev_th5 = 0, prev_d5 = 0, prev_sum8 = 0;
        int wide_settled = 0;
        int it = 0, conv = 0;
        for (it = 1; !WIDE && it <= SPKB_MAXHALF; ++it) {
            const bool odd = it & 1;   // odd: W = C^T V (column order)   even: V = C W (row order)
            double* X = odd ? W : V;
            const int rows = odd ? C : R;
            if (odd)
                spkb_product<CT, true>(keyc, minc, cntc, D, V, 0, W, 0, C, pt);
            else
                spkb_product<CT, true>(keyr, minr, cntr, D, W, 0, V, 0, R, pt);
            spk_gram(X, rows, 4, 1, sh);
            top4 = (sh.S[0] + sh.S[5]) + (sh.S[10] + sh.S[15]);
            spk_chol_factor(sh, it >= 5, trace - top4);
            if (spk_converged(top4, sh.L[11], trace, it, prev_sum, prev_delta, prev_ratio)) {
                conv = 1;
                break;
            }
            spk_orth(X, rows, 4, 1, sh);
        }
        if (WIDE) {
            // no certified gap behind the 4th value after SPKB_MAXHALF half products (clustered / slowly decaying
            // spectrum): the 8-wide fallback block of the list kernels, on the same sorted orders (two 4-column passes)
            it = it_before;
            for (int i = threadIdx.x; i < Vp; i += SPK_THREADS) {   // columns 0..3: the start block again, 4..7: noise
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    V8[(size_t)k * Vp + i] = i < R ? V[(size_t)i * 4 + k] : 0.0;
                    V8[(size_t)(4 + k) * Vp + i] = i < R ? spk_hash((unsigned)i, (unsigned)(8 + k)) : 0.0;
                }
            }
            __syncthreads();
            double th4 = 0, sum8 = 0;
            spk_wide_ritz_orth(V8, R, Vp, esh, top4, th4, sum8);   // here only as an orthonormaliser
            prev_sum = 0; prev_delta = 0; prev_ratio = 1.0;
            int wit = 1;
            for (; wit <= SPK_MAXHALF_WIDE; ++wit) {
                const bool odd = wit & 1;
                double* X = odd ? W8 : V8;
                const int rows = odd ? C : R, xcs = odd ? Wp : Vp;
                for (int cb = 0; cb < SPK_WB; cb += 4) {
                    if (odd)
                        spkb_product<CT, false>(keyc, minc, cntc, D, V8 + (size_t)cb * Vp, Vp, W8 + (size_t)cb * Wp, Wp, C, pt);
                    else
                        spkb_product<CT, false>(keyr, minr, cntr, D, W8 + (size_t)cb * Wp, Wp, V8 + (size_t)cb * Vp, Vp, R, pt);
                }
                double th5, thmin;
                spk_wide_ritz_orth(X, rows, xcs, esh, top4, th4, sum8, &th5, &thmin);
                const int verdict = spk_wide_converged(top4, th4, sum8, trace, wit, prev_sum, prev_delta, prev_ratio, th5, prev_th5, prev_d5,
                                                       thmin, wide_settled, prev_sum8);
                if (verdict) {   // 1: certified; 2: given up - flagged (status bit 0), the direct solver's (finish.hip)
                    conv = verdict == 1;
                    break;
                }
            }
            it += wit;
        }
        if (threadIdx.x == 0) {
            const double op = 1.0 - top4 / trace;
            scores[sid] = sqrt(op > 0 ? op : 0.0);
            // 4-wide kernel: bit 1 = for the wide kernel queued behind; wide kernel: bit 0 = no certificate (direct solver)
            status[sid] = (conv ? 0 : (WIDE ? 1 : 2)) | (it << 8);
        }
        __syncthreads();
    }
}

__global__ void k_iota_segments(u32* __restrict__ out, int64_t D, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = (u32)(i % D);
}
__global__ void k_segment_offsets(u32* __restrict__ off, int64_t D, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= S) off[i] = (u32)((int64_t)i * D);
}


// Common tail of the two big-table launchers: entries gathered into sorted order, slabs, the kernel, one sync.
template <typename CT>
static int big_run_kernel(sp_ctx* ctx, int64_t D, int64_t S, const u32* rr, const u32* cc, const u32* keyc, const u32* permc,
                          const u32* keyr, const u32* permr, const CT* counts, const int2* dims, int dev_cus,
                          double* scores, int* status) {
    DevBuf &minc = ctx->big[0], &minr = ctx->big[1], &cntc = ctx->big[2], &cntr = ctx->big[3], &slabs = ctx->big[4],
           &kci = ctx->big[20], &kri = ctx->big[21];
    const size_t padded = (size_t)S * (size_t)((D + SPK_THREADS - 1) / SPK_THREADS) * SPK_THREADS;
    auto cleanup = [&]() {};   // (pooled in the context)
    auto fail = [&](int code) { cleanup(); return code; };
    int rc;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(S, dev_cus));
    // row-major 4-wide V and W (4 (D + 16) doubles each), then the column-major 8-column blocks of the wide fallback
    const size_t slab_doubles = (size_t)(8 + 2 * SPK_WB) * ((size_t)D + 16);
    if ((rc = minc.ensure(padded * 4)) || (rc = minr.ensure(padded * 4)) || (rc = cntc.ensure(padded * sizeof(CT))) ||
        (rc = cntr.ensure(padded * sizeof(CT))) || (rc = kci.ensure(padded * 4)) || (rc = kri.ensure(padded * 4)) ||
        (rc = slabs.ensure((size_t)grid * slab_doubles * 8)))
        return fail(rc);
    const int64_t chunk_g = (D + SPK_THREADS - 1) / SPK_THREADS;
    const int n_ct = (int)((chunk_g + GS_T - 1) / GS_T);
    const dim3 gg((unsigned)(n_ct * (SPK_THREADS / GS_T)), (unsigned)std::min<int64_t>(S, 65535));
    hipLaunchKernelGGL(k_gather_sorted<CT>, gg, dim3(256), 0, ctx->stream, keyc, permc, rr, counts, D, n_ct, S, kci.as<u32>(),
                       minc.as<u32>(), cntc.as<CT>());   // column order: the minor index is the row
    hipLaunchKernelGGL(k_gather_sorted<CT>, gg, dim3(256), 0, ctx->stream, keyr, permr, cc, counts, D, n_ct, S, kri.as<u32>(),
                       minr.as<u32>(), cntr.as<CT>());   // row order: the minor index is the column
    const size_t lds = SPK_LDS_BYTES;   // shared structs + one or both 4-wide blocks of short sides
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sparse_big<CT, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_sparse_big<CT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        sp_set_error("big-table form: cannot reserve %zu bytes of LDS", lds);
        return fail(SP_EHIP);
    }
    hipLaunchKernelGGL((k_sparse_big<CT, false>), dim3(grid), dim3(SPK_THREADS), lds, ctx->stream, D, (int)S, rr, kci.as<u32>(),
                       minc.as<u32>(), cntc.as<CT>(), kri.as<u32>(), minr.as<u32>(), cntr.as<CT>(), counts, dims,
                       slabs.as<double>(), slab_doubles, scores, status);
    hipLaunchKernelGGL((k_sparse_big<CT, true>), dim3(grid), dim3(SPK_THREADS), lds, ctx->stream, D, (int)S, rr, kci.as<u32>(),
                       minc.as<u32>(), cntc.as<CT>(), kri.as<u32>(), minr.as<u32>(), cntr.as<CT>(), counts, dims,
                       slabs.as<double>(), slab_doubles, scores, status);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the work buffers die at return
    cleanup();
    if (e != hipSuccess) {
        sp_set_error("big-table form: %s", hipGetErrorString(e));
        return SP_EHIP;
    }
    return SP_OK;
}

// rr / cc: compact coordinates of the D patterns for each of the S splits (reindex kernel), dims: matrix sizes.
int launch_sparse_big(sp_ctx* ctx, int64_t D, int64_t S, const u32* rr, const u32* cc, const u32* counts,
                      const double* weights, const int2* dims, int dev_cus, double* scores, int* status) {
    if (S == 0) return SP_OK;
    SP_REQUIRE(D >= 1 && D < ((int64_t)1 << 31) && S * D < ((int64_t)1 << 32), SP_ELIMIT,
               "big-table form: %lld splits x %lld patterns per call is beyond the 2^32 entries one segmented sort takes",
               (long long)S, (long long)D);
    PhaseScope ps(ctx, SP_PHASE_SPARSE);
    const size_t total = (size_t)S * (size_t)D;
    DevBuf &iota = ctx->big[5], &keyc = ctx->big[6], &permc = ctx->big[7], &keyr = ctx->big[8], &permr = ctx->big[9],
           &off = ctx->big[10], &tmp = ctx->big[11];
    auto cleanup = [&]() {};   // (pooled in the context)
    auto fail = [&](int code) { cleanup(); return code; };
    int rc;
    if ((rc = iota.ensure(total * 4)) || (rc = keyc.ensure(total * 4)) || (rc = permc.ensure(total * 4)) ||
        (rc = keyr.ensure(total * 4)) || (rc = permr.ensure(total * 4)) || (rc = off.ensure((size_t)(S + 1) * 4)))
        return fail(rc);
    hipLaunchKernelGGL(k_iota_segments, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, iota.as<u32>(), D,
                       (int64_t)total);
    hipLaunchKernelGGL(k_segment_offsets, dim3((unsigned)((S + 256) / 256)), dim3(256), 0, ctx->stream, off.as<u32>(), D, (int)S);
    unsigned bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < D) ++bits;   // compact ids are < D
    // Stable sort of every split's D (compact id, pattern index) pairs by id - S independent segments of D entries
    // (radix_sort.h, round 3; rounds 1 - 2: rocprim::segmented_radix_sort_pairs).  The passes ping-pong between the named
    // buffer and a shared alternate; the first target is chosen so that the last pass lands in the named one.
    DevBuf &altk = ctx->big[12], &altv = ctx->big[13];
    if ((rc = altk.ensure(total * 4)) || (rc = altv.ensure(total * 4))) return fail(rc);
    const unsigned passes = rs_sort_passes(ctx, bits);
    auto sort_side = [&](const u32* ids, DevBuf& keyb, DevBuf& permb) -> int {
        const u32 *sk = nullptr, *sv = nullptr;
        u32 *ka = (passes & 1) ? keyb.as<u32>() : altk.as<u32>(), *kb = (passes & 1) ? altk.as<u32>() : keyb.as<u32>();
        u32 *va = (passes & 1) ? permb.as<u32>() : altv.as<u32>(), *vb = (passes & 1) ? altv.as<u32>() : permb.as<u32>();
        SP_CHECK(rs_sort<u32>(ctx, ids, ka, kb, iota.as<u32>(), va, vb, D, S, bits, tmp, &sk, &sv));
        SP_REQUIRE(sk == keyb.as<u32>() && sv == permb.as<u32>(), SP_EHIP, "big-table form: sorted lists landed in the wrong buffer");
        return SP_OK;
    };
    if ((rc = sort_side(cc, keyc, permc)) || (rc = sort_side(rr, keyr, permr))) return fail(rc);
    const int rc2 = counts ? big_run_kernel<u32>(ctx, D, S, rr, cc, keyc.as<u32>(), permc.as<u32>(), keyr.as<u32>(), permr.as<u32>(),
                                            counts, dims, dev_cus, scores, status)
                           : big_run_kernel<double>(ctx, D, S, rr, cc, keyc.as<u32>(), permc.as<u32>(), keyr.as<u32>(),
                                                    permr.as<u32>(), weights, dims, dev_cus, scores, status);
    cleanup();
    return rc2;
}

// ---- big-table form without the bitmap compaction: sides of more than 14 taxa -------------------------------------------
// The reindex kernel ranks side keys through presence bitmaps of 4^k bits, which ends at k = 14.  Here the raw side keys
// (2 bits a taxon, up to 62 bits) are sorted directly - the same stable segmented sort that orders the products - and
// the compact id of a key is the number of run heads before it.
__global__ __launch_bounds__(256) void k_side_keys(const u64* __restrict__ keys, int64_t D, int n,
                                                   const int* __restrict__ split_taxa, const int* __restrict__ split_a,
                                                   u64* __restrict__ rk, u64* __restrict__ ck) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= D) return;
    const u64 key = keys[i];
    const int* taxa = split_taxa + (size_t)s * n;
    const int a = split_a[s];
    u64 r = 0, c = 0;
    for (int t = 0; t < a; ++t) r = (r << 2) | ((key >> (2 * (n - 1 - taxa[t]))) & 3ull);
    for (int t = a; t < n; ++t) c = (c << 2) | ((key >> (2 * (n - 1 - taxa[t]))) & 3ull);
    rk[(size_t)s * D + i] = r;
    ck[(size_t)s * D + i] = c;
}

// One workgroup per split: compact id of every sorted position (= run heads before it), the same id scattered back to
// the patterns, and the number of distinct keys.
__global__ __launch_bounds__(1024) void k_compact_sorted(const u64* __restrict__ sorted_keys, const u32* __restrict__ perm,
                                                         int64_t D64, u32* __restrict__ id_sorted,
                                                         u32* __restrict__ id_by_pattern, int* __restrict__ dim_out,
                                                         int dim_stride) {
    __shared__ u32 part[1024];
    const int D = (int)D64;
    const size_t base = (size_t)blockIdx.x * D;
    const u64* k = sorted_keys + base;
    const int t = threadIdx.x;
    const int chunk = (D + 1023) / 1024;
    const int lo = min(D, t * chunk), hi = min(D, lo + chunk);
    u32 heads = 0;
    for (int j = lo; j < hi; ++j) heads += (j == 0 || k[j] != k[j - 1]) ? 1u : 0u;
    part[t] = heads;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan (Hillis-Steele)
        const u32 v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - heads;   // heads before this chunk
    for (int j = lo; j < hi; ++j) {
        run += (j == 0 || k[j] != k[j - 1]) ? 1u : 0u;
        id_sorted[base + j] = run - 1;
        id_by_pattern[base + perm[base + j]] = run - 1;
    }
    if (t == 1023) dim_out[(size_t)blockIdx.x * dim_stride] = (int)part[1023];
}

// split_taxa / split_a: host arrays of this chunk of splits.  keys: the table's pattern keys (device).
int launch_sparse_big_keys(sp_ctx* ctx, const u64* keys, int64_t D, int n, const int32_t* split_taxa, const int32_t* split_a,
                           int64_t S, const u32* counts, const double* weights, int dev_cus, double* scores, int* status) {
    if (S == 0) return SP_OK;
    SP_REQUIRE(D >= 1 && D < ((int64_t)1 << 31) && S * D < ((int64_t)1 << 32), SP_ELIMIT,
               "big-table form: %lld splits x %lld patterns per call is beyond the 2^32 entries one segmented sort takes",
               (long long)S, (long long)D);
    int max_side = 1;
    for (int64_t s = 0; s < S; ++s) max_side = std::max(max_side, std::max(split_a[s], n - split_a[s]));
    SP_REQUIRE(max_side <= 31, SP_ELIMIT, "a split side of %d taxa does not fit a 64-bit side key", max_side);
    PhaseScope ps(ctx, SP_PHASE_SPARSE);
    const size_t total = (size_t)S * (size_t)D;
    DevBuf &iota = ctx->big[5], &idc_s = ctx->big[6], &permc = ctx->big[7], &idr_s = ctx->big[8], &permr = ctx->big[9],
           &off = ctx->big[10], &tmp = ctx->big[11], &d_taxa = ctx->big[12], &d_a = ctx->big[13], &rk = ctx->big[14],
           &ck = ctx->big[15], &sk = ctx->big[16], &cc = ctx->big[17], &rr = ctx->big[18], &dims = ctx->big[19];
    auto cleanup = [&]() {};   // (pooled in the context)
    auto fail = [&](int code) { cleanup(); return code; };
    int rc;
    if ((rc = d_taxa.ensure((size_t)S * n * 4)) || (rc = d_a.ensure((size_t)S * 4)) || (rc = rk.ensure(total * 8)) ||
        (rc = ck.ensure(total * 8)) || (rc = sk.ensure(total * 8)) || (rc = iota.ensure(total * 4)) ||
        (rc = permc.ensure(total * 4)) || (rc = permr.ensure(total * 4)) || (rc = idc_s.ensure(total * 4)) ||
        (rc = idr_s.ensure(total * 4)) || (rc = cc.ensure(total * 4)) || (rc = rr.ensure(total * 4)) ||
        (rc = dims.ensure((size_t)S * sizeof(int2))) || (rc = off.ensure((size_t)(S + 1) * 4)))
        return fail(rc);
    hipError_t e = hipMemcpyAsync(d_taxa.p, split_taxa, (size_t)S * n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_a.p, split_a, (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // (pageable host arrays of the caller)
    if (e != hipSuccess) {
        sp_set_error("big-table form: %s", hipGetErrorString(e));
        return fail(SP_EHIP);
    }
    hipLaunchKernelGGL(k_side_keys, dim3((unsigned)((D + 255) / 256), (unsigned)S), dim3(256), 0, ctx->stream, keys, D, n,
                       d_taxa.as<int>(), d_a.as<int>(), rk.as<u64>(), ck.as<u64>());
    hipLaunchKernelGGL(k_iota_segments, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, iota.as<u32>(), D,
                       (int64_t)total);
    hipLaunchKernelGGL(k_segment_offsets, dim3((unsigned)((S + 256) / 256)), dim3(256), 0, ctx->stream, off.as<u32>(), D, (int)S);
    const unsigned bits = (unsigned)(2 * max_side);
    int* dimp = reinterpret_cast<int*>(dims.p);
    // stable sort of every split's (raw side key, pattern index) pairs by key: S segments of D entries (radix_sort.h)
    DevBuf &altk = ctx->big[22], &altv = ctx->big[23];   // (slots of their own: 12 / 13 are d_taxa / d_a of this function)
    if ((rc = altk.ensure(total * 8)) || (rc = altv.ensure(total * 4))) return fail(rc);
    const unsigned passes = rs_sort_passes(ctx, bits);
    auto sort_side = [&](const u64* side_keys, DevBuf& permb) -> int {
        const u64* skp = nullptr;
        const u32* svp = nullptr;
        u64 *ka = (passes & 1) ? sk.as<u64>() : altk.as<u64>(), *kb = (passes & 1) ? altk.as<u64>() : sk.as<u64>();
        u32 *va = (passes & 1) ? permb.as<u32>() : altv.as<u32>(), *vb = (passes & 1) ? altv.as<u32>() : permb.as<u32>();
        SP_CHECK(rs_sort<u64>(ctx, side_keys, ka, kb, iota.as<u32>(), va, vb, D, S, bits, tmp, &skp, &svp));
        SP_REQUIRE(skp == sk.as<u64>() && svp == permb.as<u32>(), SP_EHIP, "big-table form: sorted lists landed in the wrong buffer");
        return SP_OK;
    };
    if ((rc = sort_side(ck.as<u64>(), permc))) return fail(rc);
    hipLaunchKernelGGL(k_compact_sorted, dim3((unsigned)S), dim3(1024), 0, ctx->stream, sk.as<u64>(), permc.as<u32>(), D,
                       idc_s.as<u32>(), cc.as<u32>(), dimp + 1, 2);   // .y = columns
    if ((rc = sort_side(rk.as<u64>(), permr))) return fail(rc);
    hipLaunchKernelGGL(k_compact_sorted, dim3((unsigned)S), dim3(1024), 0, ctx->stream, sk.as<u64>(), permr.as<u32>(), D,
                       idr_s.as<u32>(), rr.as<u32>(), dimp, 2);            // .x = rows
    const int2* dims2 = reinterpret_cast<const int2*>(dims.p);
    const int rc2 = counts ? big_run_kernel<u32>(ctx, D, S, rr.as<u32>(), cc.as<u32>(), idc_s.as<u32>(), permc.as<u32>(),
                                            idr_s.as<u32>(), permr.as<u32>(), counts, dims2, dev_cus, scores, status)
                           : big_run_kernel<double>(ctx, D, S, rr.as<u32>(), cc.as<u32>(), idc_s.as<u32>(), permc.as<u32>(),
                                                    idr_s.as<u32>(), permr.as<u32>(), weights, dims2, dev_cus, scores, status);
    cleanup();
    return rc2;
}

