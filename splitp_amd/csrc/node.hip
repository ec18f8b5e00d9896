// Multi-GPU inside the library (SURVEY 8b "sp_init(n_devices) ... multi-GPU sharding + RCCL inside", 8e): ONE process drives
// the n GPUs of a node.  The candidate-split set of an alignment is sharded the way north_star prescribes - every device
// enumerates and scores the combinations rank, rank + P, ... of every size class (sp_score_all_splits_shard: an equal share
// of every cost class, no split list anywhere) on its own replica of the pattern table - and ONE ncclAllGather of the packed
// (scores, status) shards over xGMI leaves all scores on every device; device 0's copy goes to the host, un-permuted into
// the reference's all_splits order (splits.py:39-59).
// This is the C-ABI form of what splitp_amd.batch.score_all_splits(distributed=True) does with one process per GPU through
// torch.distributed (the process model bench.py uses); it exists for hosts that are not Python.  RCCL is loaded on first use
// (dlopen: the library proper links nothing but the HIP runtime), communicators come from ncclCommInitAll, the shards are
// computed by one host thread per device, the collective is issued as one group from the calling thread.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <thread>

#include "common.h"

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static int load_rccl(RcclApi& api) {
    if (api.handle) return SP_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    SP_REQUIRE(api.handle, SP_EHIP, "multi-GPU: cannot load librccl.so (%s)", dlerror());
    auto sym = [&](const char* n) { return dlsym(api.handle, n); };
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    SP_REQUIRE(api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.GetErrorString, SP_EHIP,
               "multi-GPU: librccl.so lacks an expected symbol");
    return SP_OK;
}

struct sp_node {
    int n = 0;
    bool emulated = false;   // test mode: n ranks, all on device 0, the collective replaced by device copies (no RCCL)
    RcclApi rccl;
    std::vector<sp_ctx*> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<DevBuf> send, recv;
};

#define SP_NCCL(node, call)                                                                            \
    do {                                                                                               \
        ncclResult_t r__ = (call);                                                                     \
        if (r__ != ncclSuccess) {                                                                      \
            sp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, (node)->rccl.GetErrorString(r__)); \
            return SP_EHIP;                                                                            \
        }                                                                                              \
    } while (0)

extern "C" int sp_node_destroy(sp_node* node) {
    return sp_guard("sp_node_destroy", [&]() -> int {
    if (!node) return SP_OK;
    for (int d = 0; d < (int)node->ctx.size(); ++d) {
        if (node->ctx[d]) {
            (void)hipSetDevice(node->ctx[d]->device);
            (void)hipStreamSynchronize(node->ctx[d]->stream);
        }
        if (d < (int)node->comms.size() && node->comms[d] && node->rccl.CommDestroy) (void)node->rccl.CommDestroy(node->comms[d]);
        if (d < (int)node->send.size()) node->send[d].release();
        if (d < (int)node->recv.size()) node->recv[d].release();
        if (node->ctx[d]) (void)sp_ctx_destroy(node->ctx[d]);
    }
    delete node;
    return SP_OK;
    });
}

extern "C" int sp_node_create(int n_devices, sp_node** out) {
    return sp_guard("sp_node_create", [&]() -> int {
    SP_REQUIRE(out, SP_EINVAL, "sp_node_create: out is NULL");
    const int have = sp_device_count();
    SP_REQUIRE(have > 0, SP_EHIP, "sp_node_create: no HIP device visible (this library has no CPU fallback)");
    // n_devices < 0: TEST MODE - |n_devices| ranks emulated on device 0 (every rank its own context, stream and buffers; the
    // all-gather is done with device copies): the sharding, packing and un-permuting of P > 1 ranks under test on a one-GPU box
    const bool emulated = n_devices < 0;
    if (emulated) n_devices = -n_devices;
    if (n_devices == 0) n_devices = have;
    SP_REQUIRE(emulated ? n_devices <= 64 : n_devices <= have, SP_EINVAL, "sp_node_create: %d devices asked for, %d visible", n_devices, have);
    sp_node* node = new sp_node();
    node->n = n_devices;
    node->emulated = emulated;
    node->ctx.assign((size_t)n_devices, nullptr);
    node->comms.assign((size_t)n_devices, nullptr);
    node->send.resize((size_t)n_devices);
    node->recv.resize((size_t)n_devices);
    int rc = emulated ? SP_OK : load_rccl(node->rccl);
    for (int d = 0; d < n_devices && rc == SP_OK; ++d) rc = sp_ctx_create(emulated ? 0 : d, nullptr, &node->ctx[d]);
    if (rc == SP_OK && !emulated) {
        std::vector<int> devs((size_t)n_devices);
        for (int d = 0; d < n_devices; ++d) devs[d] = d;
        const ncclResult_t r = node->rccl.CommInitAll(node->comms.data(), n_devices, devs.data());
        if (r != ncclSuccess) {
            sp_set_error("ncclCommInitAll(%d devices) failed: %s", n_devices, node->rccl.GetErrorString(r));
            rc = SP_EHIP;
        }
    }
    if (rc != SP_OK) {
        std::string keep = sp_last_error();
        (void)sp_node_destroy(node);
        sp_set_error("%s", keep.c_str());
        return rc;
    }
    *out = node;
    return SP_OK;
    });
}

extern "C" int sp_node_info(const sp_node* node, int* n_devices) {
    return sp_guard("sp_node_info", [&]() -> int {
    SP_REQUIRE(node && n_devices, SP_EINVAL, "NULL argument");
    *n_devices = node->n;
    return SP_OK;
    });
}

static unsigned long long node_binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    unsigned long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (unsigned long long)(n - k + i) / (unsigned long long)i;
    return r;
}

// Every split of the table's taxa (all_splits order), scored on the node's devices.  keys / weights / counts / D / n_taxa /
// N as for sp_alignment_create (the table is replicated: <= a few MB); method / trivial / size as for sp_score_all_splits.
extern "C" int sp_node_score_all_splits(sp_node* node, const uint64_t* keys, const double* weights, const int64_t* counts,
                                        int64_t D, int n_taxa, int64_t N, int method, int trivial, int size,
                                        int64_t* n_splits, double* scores_host, int32_t* status_host) {
    return sp_guard("sp_node_score_all_splits", [&]() -> int {
    SP_REQUIRE(node && keys && (weights || counts), SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 31 && size >= 0 && size <= n_taxa / 2, SP_EINVAL, "bad n_taxa / size");
    const int P = node->n;
    // the layout of the shards (pure arithmetic, as splitp_amd/batch.py: shard_layout)
    std::vector<int> sizes;
    if (size > 0) sizes.push_back(size);
    else for (int b = trivial ? 1 : 2; b <= n_taxa / 2; ++b) sizes.push_back(b);
    std::vector<unsigned long long> full(sizes.size());
    int64_t total = 0, per = 0;
    for (size_t q = 0; q < sizes.size(); ++q) {
        full[q] = 2 * sizes[q] == n_taxa ? node_binom(n_taxa - 1, sizes[q] - 1) : node_binom(n_taxa, sizes[q]);
        total += (int64_t)full[q];
        per += (int64_t)((full[q] + (unsigned long long)P - 1) / (unsigned long long)P);   // rank 0 holds the largest share
    }
    if (n_splits) *n_splits = total;
    if (!scores_host && !status_host) return SP_OK;
    SP_REQUIRE(D > 0, SP_EINVAL, "empty pattern table");
    const int64_t width = per + (per + 1) / 2;   // doubles per rank: scores, then the int32 status words (padded)
    std::vector<sp_alignment*> als((size_t)P, nullptr);
    std::vector<int> rcs((size_t)P, SP_OK);
    std::vector<std::string> errs((size_t)P);
    std::vector<int64_t> got((size_t)P, 0);
    auto cleanup = [&]() {
        for (int d = 0; d < P; ++d)
            if (als[d]) (void)sp_alignment_destroy(als[d]);
    };
    {   // one host thread per device: replicate the table, enumerate + score this device's shard into its send buffer
        std::vector<std::thread> th;
        for (int d = 0; d < P; ++d)
            th.emplace_back([&, d]() {
                int rc = SP_OK;
                if (hipSetDevice(node->ctx[d]->device) != hipSuccess) rc = SP_EHIP;
                if (rc == SP_OK) rc = node->send[d].ensure((size_t)width * 8);
                if (rc == SP_OK) rc = node->recv[d].ensure((size_t)width * 8 * (size_t)P);
                if (rc == SP_OK && hipMemsetAsync(node->send[d].p, 0, (size_t)width * 8, node->ctx[d]->stream) != hipSuccess) rc = SP_EHIP;
                if (rc == SP_OK) rc = sp_alignment_create(node->ctx[d], keys, weights, counts, D, n_taxa, N, &als[d]);
                if (rc == SP_OK) {
                    double* sc = node->send[d].as<double>();
                    rc = sp_score_all_splits_shard(als[d], method, trivial, size, d, P, &got[d], nullptr, sc, nullptr, sc + per);
                    if (rc == SP_ENOCONV) rc = SP_OK;   // (flagged splits travel with their status words)
                }
                if (rc != SP_OK) errs[d] = sp_last_error();   // (thread-local message: carried to the caller's thread)
                rcs[d] = rc;
            });
        for (auto& t : th) t.join();
    }
    for (int d = 0; d < P; ++d)
        if (rcs[d] != SP_OK) {
            sp_set_error("device %d: %s", d, errs[d].c_str());
            cleanup();
            return rcs[d];
        }
    if (node->emulated) {   // test mode: rank r's shard copied into slot r of every rank's receive buffer
        for (int d = 0; d < P; ++d) SP_HIP(hipStreamSynchronize(node->ctx[d]->stream));
        for (int d = 0; d < P; ++d)
            for (int r = 0; r < P; ++r)
                SP_HIP(hipMemcpyAsync(node->recv[d].as<double>() + (size_t)r * (size_t)width, node->send[r].p, (size_t)width * 8,
                                      hipMemcpyDeviceToDevice, node->ctx[d]->stream));
    } else {
    // one all-gather over xGMI (a group: this thread issues the call of every device)
    SP_NCCL(node, node->rccl.GroupStart());
    for (int d = 0; d < P; ++d) {
        const ncclResult_t r = node->rccl.AllGather(node->send[d].p, node->recv[d].p, (size_t)width, ncclDouble, node->comms[d],
                                                    node->ctx[d]->stream);
        if (r != ncclSuccess) {
            (void)node->rccl.GroupEnd();
            sp_set_error("ncclAllGather on device %d failed: %s", d, node->rccl.GetErrorString(r));
            cleanup();
            return SP_EHIP;
        }
    }
    SP_NCCL(node, node->rccl.GroupEnd());
    }
    std::vector<double> all((size_t)width * (size_t)P);
    for (int d = 0; d < P; ++d) {
        SP_HIP(hipSetDevice(node->ctx[d]->device));
        if (d == 0)
            SP_HIP(hipMemcpyAsync(all.data(), node->recv[0].p, all.size() * 8, hipMemcpyDeviceToHost, node->ctx[0]->stream));
        SP_HIP(hipStreamSynchronize(node->ctx[d]->stream));
    }
    cleanup();
    // un-permute: the j-th result of class q on rank r is split (class start) + r + j * P of all_splits
    int64_t bad = 0;
    for (int r = 0; r < P; ++r) {
        const double* sc = all.data() + (size_t)r * (size_t)width;
        const int32_t* st = reinterpret_cast<const int32_t*>(sc + per);
        int64_t at = 0, start = 0;
        for (size_t q = 0; q < sizes.size(); ++q) {
            const int64_t cnt = (int64_t)full[q];
            for (int64_t i = start + r; i < start + cnt; i += P, ++at) {
                if (scores_host) scores_host[i] = sc[at];
                if (status_host) status_host[i] = st[at];
                bad += (st[at] & 3) != 0;
            }
            start += cnt;
        }
        SP_REQUIRE(at == got[r], SP_EHIP, "device %d returned %lld scores, its shard has %lld", r, (long long)got[r], (long long)at);
    }
    if (bad) {
        sp_set_error("%lld of %lld splits are flagged (status bit 0 / 1): upper estimates", (long long)bad, (long long)total);
        return SP_ENOCONV;
    }
    return SP_OK;
    });
}
