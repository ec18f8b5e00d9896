// Gram kernel: G = C * C^T over the smaller side of every (compact) flattening matrix, fp64 MFMA.
//
// This is the first half of the replacement for LAPACK dgesdd at splitp/phylogenetics.py:281-285:
// the score only needs the four largest eigenvalues and the trace of G.
// When the matrix holds integer site counts every product and partial sum is an integer < 2^53, so
// the fp64 MFMA accumulation is exact and G is bit-reproducible regardless of summation order.
//
// Tiling (gfx950, wave64): one 256-thread workgroup (4 waves) per 64 x 64 tile of G, upper triangle
// only (mirrored on store).  Each wave owns a 32 x 32 sub-tile = 2 x 2 v_mfma_f64_16x16x4_f64
// accumulators.  K is consumed in steps of 32: the two 64 x 32 operand panels are staged through LDS
// as fp64 (converted once from u32 counts at staging time, not per MFMA), row pitch 34 doubles so the
// operand reads (16 rows x 2 k per 32-lane group, ds_read_b64) hit 32 distinct bank pairs.  The next
// panel's global loads are issued before the current panel's MFMAs (register prefetch).
#include <algorithm>

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
// staging registers as native vectors (arrays of HIP's double2 / uint4 structs are memcpy'd and can end up in scratch)
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

#define G_TILE 64
#define G_KS 32
#define G_PITCH 34  // doubles

template <typename T>
struct PanelRegs;
template <>
struct PanelRegs<u32> {
    u32x4g v[2];  // 64 rows x 8 uint4 per panel = 512 vectors / 256 threads
};
template <>
struct PanelRegs<double> {
    f64x2 v[4];  // 64 rows x 16 double2 = 1024 vectors / 256 threads
};

__device__ __forceinline__ void panel_load(PanelRegs<u32>& r, const u32* __restrict__ base, int64_t pitch, int row0,
                                           int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = threadIdx.x + i * 256;  // 0..511
        const int row = v >> 3, cv = v & 7;
        r.v[i] = *reinterpret_cast<const u32x4g*>(base + (int64_t)(row0 + row) * pitch + k0 + cv * 4);
    }
}
__device__ __forceinline__ void panel_store(const PanelRegs<u32>& r, double* s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = threadIdx.x + i * 256;
        const int row = v >> 3, cv = v & 7;
        double* d = s + row * G_PITCH + cv * 4;
        *reinterpret_cast<f64x2*>(d) = (f64x2){(double)r.v[i].x, (double)r.v[i].y};
        *reinterpret_cast<f64x2*>(d + 2) = (f64x2){(double)r.v[i].z, (double)r.v[i].w};
    }
}
__device__ __forceinline__ void panel_load(PanelRegs<double>& r, const double* __restrict__ base, int64_t pitch,
                                           int row0, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + i * 256;  // 0..1023
        const int row = v >> 4, cv = v & 15;
        r.v[i] = *reinterpret_cast<const f64x2*>(base + (int64_t)(row0 + row) * pitch + k0 + cv * 2);
    }
}
__device__ __forceinline__ void panel_store(const PanelRegs<double>& r, double* s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + i * 256;
        const int row = v >> 4, cv = v & 15;
        *reinterpret_cast<f64x2*>(s + row * G_PITCH + cv * 2) = r.v[i];
    }
}

// grid = one workgroup per work item (split, ti, tj), ti <= tj; items beyond the split's actual
// (data-dependent) row count exit at once.
template <typename T>
__global__ __launch_bounds__(256) void k_gram(const SplitDev* __restrict__ splits, const GramItem* __restrict__ items,
                                              const int2* __restrict__ dims, const T* __restrict__ mats,
                                              double* __restrict__ grams) {
    __shared__ __attribute__((aligned(16))) double sA[G_TILE * G_PITCH];
    __shared__ __attribute__((aligned(16))) double sB[G_TILE * G_PITCH];
    const GramItem it = items[blockIdx.x];
    const int sid = it.sid;
    if (sid < 0) return;  // padding item of the XCD interleave
    const SplitDev& sp = splits[sid];
    const int ti = it.ti, tj = it.tj;
    const int2 d = dims[sid];
    const int rpad = min((d.x + 63) & ~63, sp.rcap);
    if (tj * G_TILE >= rpad) return;
    const int kpad = min((d.y + 31) & ~31, sp.pitch);
    const T* __restrict__ base = mats + sp.mat_off;
    const int64_t pitch = sp.pitch;
    const bool diag = (ti == tj);
    const int lane = threadIdx.x & 63, w = sp_wave_id();
    const int wr = w >> 1, wc = w & 1;
    const int fr = lane & 15, fk = lane >> 4;

    double4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0, 0, 0, 0};

    PanelRegs<T> ra, rb;
    panel_load(ra, base, pitch, ti * G_TILE, 0);
    if (!diag) panel_load(rb, base, pitch, tj * G_TILE, 0);
    const double* pB = diag ? sA : sB;
    for (int k0 = 0; k0 < kpad; k0 += G_KS) {
        __syncthreads();  // previous panel fully consumed
        panel_store(ra, sA);
        if (!diag) panel_store(rb, sB);
        __syncthreads();
        if (k0 + G_KS < kpad) {
            panel_load(ra, base, pitch, ti * G_TILE, k0 + G_KS);
            if (!diag) panel_load(rb, base, pitch, tj * G_TILE, k0 + G_KS);
        }
        const double* a_ptr = sA + (wr * 32 + fr) * G_PITCH + fk;
        const double* b_ptr = pB + (wc * 32 + fr) * G_PITCH + fk;
#pragma unroll
        for (int kk = 0; kk < G_KS / 4; ++kk) {
            const double a0 = a_ptr[kk * 4];
            const double a1 = a_ptr[16 * G_PITCH + kk * 4];
            const double b0 = b_ptr[kk * 4];
            const double b1 = b_ptr[16 * G_PITCH + kk * 4];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // epilogue: f64 C/D layout is col = lane & 15, row = (lane >> 4) + 4 * reg
    double* __restrict__ g = grams + sp.g_off;
    const int64_t gp = sp.g_pitch;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * G_TILE + wr * 32 + m * 16 + fk + 4 * r;
                const int col = tj * G_TILE + wc * 32 + n * 16 + fr;
                const double v = acc[m][n][r];
                g[(int64_t)row * gp + col] = v;
                if (!diag) g[(int64_t)col * gp + row] = v;
            }
}

// Work lists.  Workgroups are dealt to the 8 XCDs round-robin (blocks b and b + 8 share an XCD and its
// 4 MiB L2 - observed placement, used for speed only), and every tile of a split re-reads that
// split's row panels; so the list is built as 8 interleaved streams, stream x holding all tiles of the
// splits {x, x + 8, ...} (heaviest splits first): a split's matrix is then pulled into ONE L2
// instead of all eight.  Shorter streams are padded with sid = -1 items (no-ops).
static void interleave8(const std::vector<std::vector<GramItem>>& streams, std::vector<GramItem>& out) {
    size_t longest = 0;
    for (const auto& st : streams) longest = std::max(longest, st.size());
    out.assign(longest * 8, GramItem{-1, 0, 0});
    for (int x = 0; x < 8; ++x)
        for (size_t j = 0; j < streams[x].size(); ++j) out[j * 8 + x] = streams[x][j];
}

void build_gram_items(Plan& plan) {
    std::vector<int> order(plan.splits.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    // heaviest first: a tile costs ~pitch (the K extent), so sort by rows then K
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const SplitDev &x = plan.splits[a], &y = plan.splits[b];
        if (x.rcap != y.rcap) return x.rcap > y.rcap;
        return x.pitch > y.pitch;
    });
    plan.order = order;
    std::vector<std::vector<GramItem>> gs(8), rs(8), rs_b(8);
    plan.n_order_a = 0;
    for (size_t pos = 0; pos < order.size(); ++pos) {
        const int sid = order[pos];
        const int x = (int)(pos & 7);
        const int tiles = plan.splits[sid].rcap / G_TILE;
        const bool long_side = plan.splits[sid].rcap > EIG_SMALL_ROWS;   // (order is sorted by rcap: the long sides come first)
        if (long_side) plan.n_order_a = pos + 1;
        for (int ti = 0; ti < tiles; ++ti) {
            (long_side ? rs : rs_b)[x].push_back({sid, (int16_t)ti, 0});
            for (int tj = ti; tj < tiles; ++tj) gs[x].push_back({sid, (int16_t)ti, (int16_t)tj});
        }
    }
    interleave8(gs, plan.gram_items);
    std::vector<GramItem> ra, rb2;
    interleave8(rs, ra);
    interleave8(rs_b, rb2);
    plan.n_row_a = ra.size();
    plan.row_items = ra;
    plan.row_items.insert(plan.row_items.end(), rb2.begin(), rb2.end());
    // 128 x 128 tiles of the int8 kernel (gram_i8.hip: k_gram_i8_big).  A tile costs ~its K extent, and a short side has
    // ONE tile with the longest K of all: longest K first, so that those workgroups do not form the tail of the launch.
    std::vector<int> by_k(order);
    std::stable_sort(by_k.begin(), by_k.end(), [&](int a, int b) { return plan.splits[a].pitch > plan.splits[b].pitch; });
    std::vector<std::vector<GramItem>> bs(8);
    for (size_t pos = 0; pos < by_k.size(); ++pos) {
        const int sid = by_k[pos];
        const int tiles = (plan.splits[sid].rcap + 127) / 128;
        for (int ti = 0; ti < tiles; ++ti)
            for (int tj = ti; tj < tiles; ++tj) bs[pos & 7].push_back({sid, (int16_t)ti, (int16_t)tj});
    }
    interleave8(bs, plan.gram_items_big);
}

template <typename T>
int launch_gram(sp_ctx* ctx, const SplitDev* splits_dev, const GramItem* items_dev, int64_t n_items, const int2* dims,
                const T* mats, double* grams) {
    if (n_items == 0) return SP_OK;
    PhaseScope ps(ctx, SP_PHASE_GRAM);
    hipLaunchKernelGGL(k_gram<T>, dim3((unsigned)n_items), dim3(256), 0, ctx->stream, splits_dev, items_dev, dims, mats,
                       grams);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
template int launch_gram<u32>(sp_ctx*, const SplitDev*, const GramItem*, int64_t, const int2*, const u32*, double*);
template int launch_gram<double>(sp_ctx*, const SplitDev*, const GramItem*, int64_t, const int2*, const double*,
                                 double*);
