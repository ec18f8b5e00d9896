// Exact Gram matrix of integer count matrices on the int8 matrix cores.
//
// Same job as gram.hip (G = C C^T over the smaller side, feeding the top-4 eigen step that replaces
// LAPACK gesdd at splitp/phylogenetics.py:281-285) for alignments that hold integer site counts.
// A count c < 128^NL is split into NL 7-bit limbs, c = sum_l 128^l c_l, stored as NL int8 planes;
//     G = sum_{a,b} 128^(a+b) * (C_a C_b^T)
// and every C_a C_b^T is an int8 x int8 -> int32 MFMA product (v_mfma_i32_32x32x32_i8, 64x the fp64
// MFMA rate, so even NL^2 = 4 products are ~16x cheaper than one fp64 product).  The int32 partial
// sums are exact (127^2 * K < 2^31 for K < 133k; longer K is flushed into fp64 every 32768 columns),
// the limb recombination is exact in fp64 (< 2^53), so G is bit-identical to the fp64 kernel's.
//
// Tiling: one 256-thread workgroup per 64 x 64 upper-triangle tile of G; wave (wr, wc) owns a 32 x 32
// sub-tile = one 32x32x32 MFMA per limb pair per 32-wide k-step.  K is consumed 128 bytes at a time:
// the NL planes of both 64-row panels go global -> registers -> LDS (16-byte accesses; LDS row pitch
// 144 B so the ds_read_b128 operand reads of 16 different rows hit 64 distinct banks).  Both operands
// are fetched with the same lane -> k rule, so the product does not depend on the instruction's
// internal k ordering; only the (dtype-independent) C/D layout matters.
#include <algorithm>

#include "common.h"

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16_t_v __attribute__((ext_vector_type(16)));
// staging registers are NATIVE vectors: an array of HIP's uint4 struct is copied with memcpy and stayed in scratch memory
// (144 - 224 bytes per lane written and read back per K step: the round-1 kernel's real bound)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GI_TILE 64
#define GI_KS 128           // bytes of K per step
#define GI_PITCH 144        // LDS row pitch in bytes
#define GI_FLUSH 32768      // columns between int32 -> fp64 flushes

// GT = int (every entry of G is known to fit: max_count * N < 2^31) or double.
// (launch bounds: at least GI_WAVES_PER_SIMD workgroups' worth of waves per SIMD - left to itself the compiler unrolled
// the k loop into 332 registers + scratch, i.e. ONE workgroup per CU, and the kernel sat at 7 % MFMA busy waiting for
// its own loads: profiles/r02_pmc_dense_route_before.json)
#ifndef GI_WAVES_PER_SIMD
#define GI_WAVES_PER_SIMD 3
#endif
template <int NL, typename GT>
__global__ __launch_bounds__(256, GI_WAVES_PER_SIMD) void k_gram_i8(const SplitDev* __restrict__ splits,
                                                 const GramItem* __restrict__ items,
                                                 const int2* __restrict__ dims, const uint8_t* __restrict__ mats,
                                                 GT* __restrict__ grams) {
    // one LDS arena: operand panels during the K loop, then the 64 x 64 output tile for the epilogue
    constexpr int PANEL = GI_TILE * GI_PITCH;
    constexpr int STAGE_BYTES = 2 * NL * PANEL;
    constexpr int TILE_BYTES = GI_TILE * (GI_TILE + 1) * (int)sizeof(GT);
    __shared__ __attribute__((aligned(16))) uint8_t arena[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    uint8_t(*sA)[PANEL] = reinterpret_cast<uint8_t(*)[PANEL]>(arena);
    uint8_t(*sB)[PANEL] = reinterpret_cast<uint8_t(*)[PANEL]>(arena + NL * PANEL);
    const GramItem it = items[blockIdx.x];
    const int sid = it.sid;
    if (sid < 0) return;  // padding item of the XCD interleave
    const SplitDev& sp = splits[sid];
    const int ti = it.ti, tj = it.tj;
    const int2 d = dims[sid];
    const int rpad = min((d.x + 63) & ~63, sp.rcap);
    if (tj * GI_TILE >= rpad) return;
    const int kpad = min((d.y + GI_KS - 1) & ~(GI_KS - 1), sp.pitch);
    const int64_t pitch = sp.pitch;                       // bytes
    const int64_t plane = (int64_t)sp.rcap * pitch;       // bytes per limb plane
    const uint8_t* __restrict__ base = mats + sp.mat_off;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int wr = w >> 1, wc = w & 1;
    const int fr = lane & 31, fh = lane >> 5;

    // staging map: a panel limb is 64 rows x 128 B = 512 x 16 B -> 2 vectors per thread
    const int v0 = threadIdx.x, v1 = threadIdx.x + 256;
    const int r0 = v0 >> 3, c0 = (v0 & 7) * 16, r1 = v1 >> 3, c1 = (v1 & 7) * 16;

    // one int32 accumulator set per limb weight a + b (products of equal weight share a set: NL = 2 needs 3 sets, not
    // 4; NL = 3 needs 5, not 9); NL products of 127^2 * GI_FLUSH columns stay below 2^31
    constexpr int NS = 2 * NL - 1;
    static_assert((long long)NL * 127 * 127 * GI_FLUSH < (1ll << 31), "int32 partial sums");
    int16_t_v acc[NS];
    long long facc[16];
#pragma unroll
    for (int w2 = 0; w2 < NS; ++w2)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[w2][e] = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) facc[e] = 0;

    u32x4 ra[NL][2], rb[NL][2];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const uint8_t* pa = base + l * plane + (int64_t)(ti * GI_TILE) * pitch + k0;
            ra[l][0] = *reinterpret_cast<const u32x4*>(pa + (int64_t)r0 * pitch + c0);
            ra[l][1] = *reinterpret_cast<const u32x4*>(pa + (int64_t)r1 * pitch + c1);
            if (!diag) {
                const uint8_t* pb = base + l * plane + (int64_t)(tj * GI_TILE) * pitch + k0;
                rb[l][0] = *reinterpret_cast<const u32x4*>(pb + (int64_t)r0 * pitch + c0);
                rb[l][1] = *reinterpret_cast<const u32x4*>(pb + (int64_t)r1 * pitch + c1);
            }
        }
    };
    gload(0);
    for (int kc = 0; kc < kpad; kc += GI_FLUSH) {           // int32 partial sums are flushed per chunk
        const int kend = min(kpad, kc + GI_FLUSH);
        for (int k0 = kc; k0 < kend; k0 += GI_KS) {
            __syncthreads();  // previous panel fully consumed
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                *reinterpret_cast<u32x4*>(&sA[l][r0 * GI_PITCH + c0]) = ra[l][0];
                *reinterpret_cast<u32x4*>(&sA[l][r1 * GI_PITCH + c1]) = ra[l][1];
                if (!diag) {
                    *reinterpret_cast<u32x4*>(&sB[l][r0 * GI_PITCH + c0]) = rb[l][0];
                    *reinterpret_cast<u32x4*>(&sB[l][r1 * GI_PITCH + c1]) = rb[l][1];
                }
            }
            __syncthreads();
            if (k0 + GI_KS < kpad) gload(k0 + GI_KS);
#pragma unroll
            for (int kk = 0; kk < GI_KS / 32; ++kk) {
                int4_t fa[NL], fb[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    fa[l] = *reinterpret_cast<const int4_t*>(&sA[l][(wr * 32 + fr) * GI_PITCH + kk * 32 + fh * 16]);
                    fb[l] = diag
                                ? *reinterpret_cast<const int4_t*>(&sA[l][(wc * 32 + fr) * GI_PITCH + kk * 32 + fh * 16])
                                : *reinterpret_cast<const int4_t*>(&sB[l][(wc * 32 + fr) * GI_PITCH + kk * 32 + fh * 16]);
                }
#pragma unroll
                for (int a = 0; a < NL; ++a)
#pragma unroll
                    for (int b = 0; b < NL; ++b)
                        acc[a + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a + b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int w2 = 0; w2 < NS; ++w2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                facc[e] += (long long)acc[w2][e] << (7 * w2);
                acc[w2][e] = 0;
            }
        }
    }
    // epilogue: 32x32 C/D layout is col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // The tile goes through LDS so that both G[ti-block][tj-block] and its mirror G[tj-block][ti-block]
    // leave as full 64-element rows (coalesced): writing the mirror straight from the accumulators is
    // one scattered 8-byte store per element (measured: 4.5x write amplification).
    __syncthreads();
    GT* tile = reinterpret_cast<GT*>(arena);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int col = wc * 32 + fr;
        tile[row * (GI_TILE + 1) + col] = (GT)facc[e];
    }
    __syncthreads();
    GT* __restrict__ g = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
    for (int e = threadIdx.x; e < GI_TILE * GI_TILE; e += 256) {
        const int r = e >> 6, c = e & 63;
        g[(int64_t)(ti * GI_TILE + r) * gp + tj * GI_TILE + c] = tile[r * (GI_TILE + 1) + c];
        if (!diag) g[(int64_t)(tj * GI_TILE + r) * gp + ti * GI_TILE + c] = tile[c * (GI_TILE + 1) + r];
    }
}


// ---- 128 x 128 tiles (round 2) ---------------------------------------------------------------------------------------
// The 64 x 64 kernel above moves 2 x 64 panel rows per 64 x 64 outputs through L2 -> LDS: at config 2 that is 6.7 GB of
// panel reads per launch for 0.9 GB of matrices (105 M L2 requests, profiles/r02_pmc_dense_route.json) and the matrix
// cores sat at 7 % busy whatever the occupancy.  Here one 512-thread workgroup owns a 128 x 128 tile (half the panel
// bytes per output), its 8 waves in a 2 x 4 grid of 64 x 32 sub-tiles (2 MFMA tiles, 3 accumulator sets each: 96
// registers, so two waves share a SIMD and one wave's LDS reads hide behind the other's MFMAs).  K goes 128 bytes a step
// through the same register-staged, 144-byte-pitch LDS image (conflict-free ds_read_b128).  Waves whose sub-tile lies
// wholly below the diagonal of a diagonal tile, or wholly beyond the rows in use, stage and synchronise but issue no
// MFMA.  Rows >= rpad are never read (k_zero_i8 clears rpad rows only: the padding of a 128-row tile is loaded as
// zeros).  Epilogue: every wave stores its sub-tiles straight from the accumulators (two 128-byte row segments per
// instruction) and the mirror image through a wave-private 32 x 33 LDS transpose (rows of 32 again).
// One int32 chunk only: the launcher keeps this kernel to K <= GB_KMAX bytes (2 products of 127^2 per column and set).
#define GB_TILE 128
#define GB_KMAX 65536
template <int NL, typename GT>
__global__ __launch_bounds__(512, 2) void k_gram_i8_big(const SplitDev* __restrict__ splits,
                                                        const GramItem* __restrict__ items,
                                                        const int2* __restrict__ dims, const uint8_t* __restrict__ mats,
                                                        GT* __restrict__ grams) {
    constexpr int PANEL = GB_TILE * GI_PITCH;
    constexpr int STAGE_BYTES = 2 * NL * PANEL;
    constexpr int SCRATCH_BYTES = 8 * 32 * 33 * (int)sizeof(GT);
    static_assert(2ll * 127 * 127 * GB_KMAX < (1ll << 31), "int32 partial sums");
    __shared__ __attribute__((aligned(16))) uint8_t arena[STAGE_BYTES > SCRATCH_BYTES ? STAGE_BYTES : SCRATCH_BYTES];
    uint8_t(*sA)[PANEL] = reinterpret_cast<uint8_t(*)[PANEL]>(arena);
    uint8_t(*sB)[PANEL] = reinterpret_cast<uint8_t(*)[PANEL]>(arena + NL * PANEL);
    const GramItem it = items[blockIdx.x];
    const int sid = it.sid;
    if (sid < 0) return;  // padding item of the XCD interleave
    const SplitDev& sp = splits[sid];
    const int ti = it.ti, tj = it.tj;
    const int2 d = dims[sid];
    const int rpad = min((d.x + 63) & ~63, sp.rcap);
    if (tj * GB_TILE >= rpad) return;
    const int kpad = min((d.y + GI_KS - 1) & ~(GI_KS - 1), sp.pitch);
    const int64_t pitch = sp.pitch;                       // bytes
    const int64_t plane = (int64_t)sp.rcap * pitch;       // bytes per limb plane
    const uint8_t* __restrict__ base = mats + sp.mat_off;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int wr = w >> 2, wc = w & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int row0 = ti * GB_TILE + wr * 64, col0 = tj * GB_TILE + wc * 32;   // this wave's 64 x 32 sub-tile of G
    // nothing to compute: sub-tile wholly below the diagonal (its mirror is computed by another wave) or beyond the rows
    const bool idle = (diag && wc * 32 + 32 <= wr * 64) || row0 >= rpad || col0 >= rpad;

    // staging map: a panel limb is 128 rows x 128 B = 1024 x 16 B -> 2 vectors per thread
    const int r0 = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 16, r1 = r0 + 64;
    const bool a0 = ti * GB_TILE + r0 < rpad, a1 = ti * GB_TILE + r1 < rpad;
    const bool b0 = !diag && tj * GB_TILE + r0 < rpad, b1 = !diag && tj * GB_TILE + r1 < rpad;

    constexpr int NS = 2 * NL - 1;
    int16_t_v acc[2][NS];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int w2 = 0; w2 < NS; ++w2)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][w2][e] = 0;

    u32x4 ra[NL][2], rb[NL][2];
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const uint8_t* pa = base + l * plane + (int64_t)(ti * GB_TILE) * pitch + k0 + c0;
            ra[l][0] = a0 ? *reinterpret_cast<const u32x4*>(pa + (int64_t)r0 * pitch) : z4;
            ra[l][1] = a1 ? *reinterpret_cast<const u32x4*>(pa + (int64_t)r1 * pitch) : z4;
            if (!diag) {
                const uint8_t* pb = base + l * plane + (int64_t)(tj * GB_TILE) * pitch + k0 + c0;
                rb[l][0] = b0 ? *reinterpret_cast<const u32x4*>(pb + (int64_t)r0 * pitch) : z4;
                rb[l][1] = b1 ? *reinterpret_cast<const u32x4*>(pb + (int64_t)r1 * pitch) : z4;
            }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < kpad; k0 += GI_KS) {
        __syncthreads();  // previous panel fully consumed
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            *reinterpret_cast<u32x4*>(&sA[l][r0 * GI_PITCH + c0]) = ra[l][0];
            *reinterpret_cast<u32x4*>(&sA[l][r1 * GI_PITCH + c0]) = ra[l][1];
            if (!diag) {
                *reinterpret_cast<u32x4*>(&sB[l][r0 * GI_PITCH + c0]) = rb[l][0];
                *reinterpret_cast<u32x4*>(&sB[l][r1 * GI_PITCH + c0]) = rb[l][1];
            }
        }
        __syncthreads();
        if (k0 + GI_KS < kpad) gload(k0 + GI_KS);
        if (!idle) {
            uint8_t(*sBB)[PANEL] = diag ? sA : sB;
#pragma unroll
            for (int kk = 0; kk < GI_KS / 32; ++kk) {
                int4_t fa[2][NL], fb[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    fa[0][l] = *reinterpret_cast<const int4_t*>(&sA[l][(wr * 64 + fr) * GI_PITCH + kk * 32 + fh * 16]);
                    fa[1][l] = *reinterpret_cast<const int4_t*>(&sA[l][(wr * 64 + 32 + fr) * GI_PITCH + kk * 32 + fh * 16]);
                    fb[l] = *reinterpret_cast<const int4_t*>(&sBB[l][(wc * 32 + fr) * GI_PITCH + kk * 32 + fh * 16]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int a = 0; a < NL; ++a)
#pragma unroll
                        for (int b = 0; b < NL; ++b)
                            acc[m][a + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[m][a], fb[b], acc[m][a + b], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // the arena becomes the transpose scratch
    if (idle) return;
    GT* scratch = reinterpret_cast<GT*>(arena) + w * (32 * 33);
    GT* __restrict__ g = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int rb0 = row0 + m * 32;                    // first row of this 32 x 32 MFMA tile
        if (rb0 >= rpad) continue;
        if (diag && col0 + 32 <= rb0) continue;           // wholly below the diagonal: written as a mirror
        // 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            long long v = 0;
#pragma unroll
            for (int w2 = 0; w2 < NS; ++w2) v += (long long)acc[m][w2][e] << (7 * w2);
            const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
            const GT gv = (GT)v;
            g[(int64_t)(rb0 + rl) * gp + col0 + fr] = gv;
            scratch[rl * 33 + fr] = gv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (rb0 != col0) {                                // mirror: G[col0 + c][rb0 + r], rows of 32 consecutive r
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cl = (e & 3) + 8 * (e >> 2) + 4 * fh;
                g[(int64_t)(col0 + cl) * gp + rb0 + fr] = scratch[fr * 33 + cl];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <typename GT>
static int launch_gram_i8_big_t(sp_ctx* ctx, int nl, const SplitDev* splits_dev, const GramItem* items_dev,
                                int64_t n_items, const int2* dims, const uint8_t* mats, GT* grams) {
    switch (nl) {
        case 1:
            hipLaunchKernelGGL((k_gram_i8_big<1, GT>), dim3((unsigned)n_items), dim3(512), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        case 2:
            hipLaunchKernelGGL((k_gram_i8_big<2, GT>), dim3((unsigned)n_items), dim3(512), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        case 3:
            hipLaunchKernelGGL((k_gram_i8_big<3, GT>), dim3((unsigned)n_items), dim3(512), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        default:
            sp_set_error("launch_gram_i8_big: unsupported limb count %d", nl);
            return SP_EINVAL;
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// 128 x 128-tile form; the caller has checked that every split's K extent is <= GB_KMAX bytes.
int launch_gram_i8_big(sp_ctx* ctx, int nl, bool g_i32, const SplitDev* splits_dev, const GramItem* items_dev,
                       int64_t n_items, const int2* dims, const uint8_t* mats, void* grams) {
    if (n_items == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_GRAM);
    if (g_i32) return launch_gram_i8_big_t<int>(ctx, nl, splits_dev, items_dev, n_items, dims, mats, (int*)grams);
    return launch_gram_i8_big_t<double>(ctx, nl, splits_dev, items_dev, n_items, dims, mats, (double*)grams);
}

template <typename GT>
static int launch_gram_i8_t(sp_ctx* ctx, int nl, const SplitDev* splits_dev, const GramItem* items_dev, int64_t n_items,
                            const int2* dims, const uint8_t* mats, GT* grams) {
    switch (nl) {
        case 1:
            hipLaunchKernelGGL((k_gram_i8<1, GT>), dim3((unsigned)n_items), dim3(256), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        case 2:
            hipLaunchKernelGGL((k_gram_i8<2, GT>), dim3((unsigned)n_items), dim3(256), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        case 3:
            hipLaunchKernelGGL((k_gram_i8<3, GT>), dim3((unsigned)n_items), dim3(256), 0, ctx->stream, splits_dev,
                               items_dev, dims, mats, grams);
            break;
        default:
            sp_set_error("launch_gram_i8: unsupported limb count %d", nl);
            return SP_EINVAL;
    }
    SP_HIP(hipGetLastError());
    return SP_OK;
}

int launch_gram_i8(sp_ctx* ctx, int nl, bool g_i32, const SplitDev* splits_dev, const GramItem* items_dev,
                   int64_t n_items, const int2* dims, const uint8_t* mats, void* grams) {
    if (n_items == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_GRAM);
    if (g_i32) return launch_gram_i8_t<int>(ctx, nl, splits_dev, items_dev, n_items, dims, mats, (int*)grams);
    return launch_gram_i8_t<double>(ctx, nl, splits_dev, items_dev, n_items, dims, mats, (double*)grams);
}
