"""Batched scoring: many candidate splits of one alignment in one device pass, sharded across
GPUs with an all-gather of the scores.

This is the device-resident form of the reference's README loop (README.md:36-41):
    for split in splits: split_score(flattening(split, alignment, FlatFormat.reduced))
and of one round of erickson_SVD (splitp/phylogenetics.py:126-140).  The candidate-split set is
dealt to ranks by cost class; each rank scores its shard with libsplitp_hip.so; one
all_gather (RCCL when the process group's backend is "nccl") returns every score to every rank."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import as_device_alignment, normalise_split, resolve_split
from .enums import Method


def encode_splits(splits, table, n_taxa):
    """-> (split_taxa int32 [S, n], split_a int32 [S]) in the C ABI's layout."""
    s_count = len(splits)
    taxa = getattr(table, "taxa", None)
    if s_count:   # fast path: one flat comprehension instead of a loop body per split (3x on 2035 splits)
        try:
            norm = [normalise_split(sp) for sp in splits]
            t0 = taxa if taxa is not None else sorted(set.union(*map(set, norm[0])))
            lookup = {x: j for j, x in enumerate(t0)}
            a_fast = np.fromiter((len(sp[0]) for sp in norm), dtype=np.int32, count=s_count)
            b_fast = np.fromiter((len(sp[1]) for sp in norm), dtype=np.int32, count=s_count)
            if np.all(a_fast + b_fast == n_taxa) and all(len(sp) == 2 for sp in norm):
                flat = [lookup[x] for sp in norm for side in sp for x in side]
                return np.array(flat, dtype=np.int32).reshape(s_count, n_taxa), a_fast
        except (KeyError, TypeError, IndexError):
            pass      # fall through to the loop below, which names the offending split
    taxa_arr = np.empty((s_count, n_taxa), dtype=np.int32)
    a_arr = np.empty(s_count, dtype=np.int32)
    where = None
    for i, sp in enumerate(splits):
        sp = normalise_split(sp)
        if where is None:
            t = taxa if taxa is not None else sorted(set.union(*map(set, sp)))
            where = {x: j for j, x in enumerate(t)}
        if len(sp[0]) + len(sp[1]) != n_taxa:
            raise ValueError(f"split {sp} does not cover all {n_taxa} taxa of the table")
        a_arr[i] = len(sp[0])
        taxa_arr[i, : len(sp[0])] = [where[x] for x in sp[0]]
        taxa_arr[i, len(sp[0]):] = [where[x] for x in sp[1]]
    return taxa_arr, a_arr


def encode_all_splits(n_taxa, trivial=False, size=None):
    """(split_taxa, split_a) of `all_splits(taxa)` - same splits, same order (reference splits.py:39-59) - built with
    NumPy instead of one Python tuple pair per split: the 32 751 splits of a 16-taxon alignment take 18 ms instead of
    the 170 ms of `encode_splits(list(all_splits(taxa)))`, the 524 267 of 20 taxa 1.2 s instead of ~3 s.  Taxa are positions 0 .. n_taxa-1 of the table's taxon list."""
    from itertools import chain, combinations
    from math import comb

    n = int(n_taxa)
    sizes = [size] if size is not None else list(range(1 if trivial else 2, n // 2 + 1))
    taxa_blocks, a_blocks = [], []
    everyone = np.arange(n, dtype=np.int32)
    for bal in sizes:
        even = 2 * bal == n
        if even:   # taxon 0 plus every (bal-1)-subset of the others
            cnt = comb(n - 1, bal - 1)
            rest = np.fromiter(chain.from_iterable(combinations(range(1, n), bal - 1)), dtype=np.int32,
                               count=cnt * (bal - 1)).reshape(cnt, bal - 1)
            chosen = np.concatenate([np.zeros((cnt, 1), dtype=np.int32), rest], axis=1)
        else:
            cnt = comb(n, bal)
            chosen = np.fromiter(chain.from_iterable(combinations(range(n), bal)), dtype=np.int32,
                                 count=cnt * bal).reshape(cnt, bal)
        member = np.zeros((cnt, n), dtype=bool)
        member[np.arange(cnt)[:, None], chosen] = True
        # the side holding taxon 0 goes first (splits.py:55-56); both sides in taxon order
        first = member == member[:, :1]
        a = first.sum(axis=1).astype(np.int32)
        c_first = np.cumsum(first, axis=1, dtype=np.int32)
        # position of taxon t in the row: its rank inside its side, the second side shifted behind the first
        pos = np.where(first, c_first - 1, a[:, None] + (everyone[None, :] - c_first))
        out = np.empty((cnt, n), dtype=np.int32)
        np.put_along_axis(out, pos, np.broadcast_to(everyone, (cnt, n)), axis=1)
        taxa_blocks.append(out)
        a_blocks.append(a)
    if not taxa_blocks:
        return np.zeros((0, n), dtype=np.int32), np.zeros(0, dtype=np.int32)
    return np.ascontiguousarray(np.concatenate(taxa_blocks)), np.ascontiguousarray(np.concatenate(a_blocks))


def split_costs(split_a, n_taxa, method):
    """Relative cost of a split (for balanced sharding): the Gram over the smaller side costs
    ~ 4^k * D on the flattening route; the subflattening route's eigenproblem ~ (3k+1)^3."""
    k = np.minimum(split_a, n_taxa - split_a).astype(np.int64)
    if method != _lib.SP_METHOD_SUBFLATTENING:
        return (4.0 ** k)
    return (3.0 * k + 1.0) ** 3


def shard_indices(costs, world_size):
    """Deal splits to ranks: sort by (cost class desc, original index) and deal round-robin, so
    every rank receives an equal share of every cost class.  Returns a list of index arrays
    (deterministic; identical on every rank)."""
    order = np.lexsort((np.arange(len(costs)), -np.asarray(costs)))
    return [order[r::world_size] for r in range(world_size)]


def _method_code(method, route="auto"):
    name = getattr(method, "name", method)
    if name in ("flattening", Method.flattening):
        return {"auto": _lib.SP_METHOD_FLATTENING, "dense": _lib.SP_METHOD_FLATTENING_DENSE,
                "sparse": _lib.SP_METHOD_FLATTENING_SPARSE}[route]
    if name in ("subflattening", Method.subflattening):
        return _lib.SP_METHOD_SUBFLATTENING
    if name in ("mutual_information", Method.mutual_information):
        return _lib.SP_METHOD_MUTUAL_INFORMATION
    raise ValueError(f"unsupported method {method!r} (flattening, subflattening or mutual_information)")


def warn_unconverged(status, stacklevel=3):
    """RuntimeWarning when any status word carries bit 0 (the C ABI's SP_ENOCONV: score written, upper estimate)."""
    status = np.asarray(status)
    bad = int(np.count_nonzero(status & 1))
    if bad:
        import warnings

        warnings.warn(f"{bad} of {status.size} splits hit the iteration cap of their eigen-solver: "
                      "their scores are upper estimates (status bit 0)", RuntimeWarning, stacklevel=stacklevel)
    return bad


def score_encoded(al, split_taxa, split_a, method_code, scores_dev_ptr=None, want_host=True):
    """Score already-encoded splits on this process's GPU.  Returns (scores, status) host arrays
    when want_host, else enqueues only (scores land in scores_dev_ptr).  The library reports unconverged splits
    with SP_ENOCONV (scores still written): that becomes a RuntimeWarning here, the status words say which."""
    n = len(split_a)
    split_taxa = np.ascontiguousarray(split_taxa, dtype=np.int32)
    split_a = np.ascontiguousarray(split_a, dtype=np.int32)
    scores = np.empty(n, dtype=np.float64) if want_host else None
    status = np.zeros(n, dtype=np.int32) if want_host else None
    _lib.check(al.ctx._lib.sp_score_splits(
        al.handle, _lib._ptr(split_taxa, C.c_int32), _lib._ptr(split_a, C.c_int32), n, method_code,
        _lib._ptr(scores, C.c_double), C.c_void_p(scores_dev_ptr) if scores_dev_ptr else None,
        _lib._ptr(status, C.c_int32)), allow_noconv=True)
    if want_host:
        warn_unconverged(status)
    return scores, status


class SplitPlan:
    """A candidate-split list as an immutable device object (sp_plan): planned and uploaded once, shared read-only by
    any number of contexts (lanes) and alignments of the device."""

    def __init__(self, ctx, split_taxa, split_a, n_taxa):
        import weakref

        self.split_taxa = np.ascontiguousarray(split_taxa, dtype=np.int32)
        self.split_a = np.ascontiguousarray(split_a, dtype=np.int32)
        self.n_taxa = int(n_taxa)
        self.n_splits = len(self.split_a)
        self.handle = C.c_void_p()
        lib = ctx._lib
        _lib.check(lib.sp_plan_create(ctx.handle, self.n_taxa, _lib._ptr(self.split_taxa, C.c_int32),
                                      _lib._ptr(self.split_a, C.c_int32), self.n_splits, C.byref(self.handle)))
        self._fin = weakref.finalize(self, lib.sp_plan_release, self.handle)

    @classmethod
    def from_splits(cls, splits, table):
        al = as_device_alignment(table)
        taxa_arr, a_arr = encode_splits(list(splits), table, al.n_taxa)
        return cls(al.ctx, taxa_arr, a_arr, al.n_taxa)


def score_plan_async(lane_ctx, als, plan, scores_dev_ptr, status_dev_ptr, _handles=None):
    """sp_score_plan_async: every split of `plan` for each alignment in `als`, enqueued on the lane's stream, complete on
    the device (hand-back chain included); results alignment-major in the given device buffers."""
    arr = _handles if _handles is not None else (C.c_void_p * len(als))(*[a.handle.value for a in als])
    _lib.check(lane_ctx._lib.sp_score_plan_async(lane_ctx.handle, arr, len(als), plan.handle,
                                                 C.c_void_p(scores_dev_ptr), C.c_void_p(status_dev_ptr)))


def score_plan_steps(lane_ctx, als, plan, n_steps, scores_dev_ptr, scores_step_bytes, status_dev_ptr, status_step_bytes,
                     _handles=None):
    """sp_score_plan_steps: `n_steps` complete passes of score_plan_async from one host call; pass s writes its scores /
    status words `s * step_bytes` behind the given device pointers."""
    arr = _handles if _handles is not None else (C.c_void_p * len(als))(*[a.handle.value for a in als])
    _lib.check(lane_ctx._lib.sp_score_plan_steps(lane_ctx.handle, arr, len(als), plan.handle, int(n_steps),
                                                 C.c_void_p(scores_dev_ptr), int(scores_step_bytes),
                                                 C.c_void_p(status_dev_ptr), int(status_step_bytes)))


def score_encoded_async(al, split_taxa, split_a, method_code, scores_dev_ptr, status_dev_ptr):
    """Enqueue only (no host synchronisation): scores and status land in the given device buffers.  On the sparse
    route the hand-back chain runs on the device; on every flattening route a split whose eigen-solver found no
    certificate comes back flagged (status bit 0 / 1) - finish_async re-scores those with the direct solver."""
    split_taxa = np.ascontiguousarray(split_taxa, dtype=np.int32)
    split_a = np.ascontiguousarray(split_a, dtype=np.int32)
    _lib.check(al.ctx._lib.sp_score_splits_async(
        al.handle, _lib._ptr(split_taxa, C.c_int32), _lib._ptr(split_a, C.c_int32), len(split_a), method_code,
        C.c_void_p(scores_dev_ptr), C.c_void_p(status_dev_ptr)))


def score_encoded_multi_async(als, split_taxa, split_a, scores_dev_ptr, status_dev_ptr):
    """Several alignments (same taxa / split list) in one device pass of the sparse route; results alignment-major."""
    split_taxa = np.ascontiguousarray(split_taxa, dtype=np.int32)
    split_a = np.ascontiguousarray(split_a, dtype=np.int32)
    arr = (C.c_void_p * len(als))(*[a.handle.value for a in als])
    _lib.check(als[0].ctx._lib.sp_score_splits_multi_async(
        arr, len(als), _lib._ptr(split_taxa, C.c_int32), _lib._ptr(split_a, C.c_int32), len(split_a),
        C.c_void_p(scores_dev_ptr), C.c_void_p(status_dev_ptr)))


def finish_async(al, split_taxa, split_a, scores_host, status_host):
    """Host step behind the asynchronous form: the device chain certifies a score or flags it (status bit 0: the 8-wide
    block found no certificate - a spectrum without a gap behind its 4th value).  Flagged splits are re-scored by the
    library's direct solver (sp_finish_flagged: Householder tridiagonalisation + Sturm counts on the split's Gram matrix,
    no stop rule) and patched in place; their status word then has bit 2 set and bits 0 / 1 clear.
    scores_host / status_host are NumPy arrays of the fetched results; returns how many splits were finished."""
    status_host = np.asarray(status_host)
    if not np.any(status_host & 3):
        return 0
    sc = np.ascontiguousarray(scores_host, dtype=np.float64)
    st = np.ascontiguousarray(status_host, dtype=np.int32)
    split_taxa = np.ascontiguousarray(split_taxa, dtype=np.int32)
    split_a = np.ascontiguousarray(split_a, dtype=np.int32)
    done = C.c_int64(0)
    _lib.check(al.ctx._lib.sp_finish_flagged(al.handle, _lib._ptr(split_taxa, C.c_int32), _lib._ptr(split_a, C.c_int32),
                                             len(split_a), _lib._ptr(sc, C.c_double), _lib._ptr(st, C.c_int32),
                                             C.byref(done)))
    if sc is not scores_host:
        scores_host[...] = sc
    if st is not status_host:
        status_host[...] = st
    return int(done.value)


def _send_buffer(width):
    """Zeroed device buffer for a collective's send side, COMPLETE before it is returned.  The library's kernels run on the
    context's stream; when torch's current stream is the legacy default stream the context owns a private one, so neither
    torch's fill kernel nor the collective is ordered with them by the stream alone: the fill is waited for here, the
    kernels by a context synchronise before the collective (found by the world-size-1 nccl test of round 3: the 0.7 ms
    subflattening pass of a 16-taxon table had its results overwritten by the late fill - all scores 0)."""
    import torch

    send = torch.zeros(width, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.current_stream().synchronize()
    return send


def packed_width(per):
    """doubles per rank in the exchange buffer: `per` scores, then `per` int32 status words (padded to a double)."""
    return per + (per + 1) // 2


def gather_scores(local_scores, shards, n_total, group=None, device_tensor=None, local_status=None,
                  return_status=False):
    """All-gather the per-rank score shards and un-permute them into split order.

    local_scores / local_status: this rank's shard (len(shards[rank])).  ONE collective moves both: each rank sends
    packed_width(per) doubles - its scores, then its int32 status words - so the un-converged flag (status bit 0)
    reaches every rank with the scores.  With the "nccl" backend (RCCL over xGMI) that is one all_gather_into_tensor
    on the GPU (device_tensor = the already packed device buffer); with "gloo" (CPU tests) the same call on host
    tensors.  Returns scores, or (scores, status) with return_status."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = max(len(s) for s in shards)
    width = packed_width(per)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    if device_tensor is not None:
        send = device_tensor
    else:
        host = np.zeros(width, dtype=np.float64)
        mine = len(shards[rank])
        host[:mine] = np.asarray(local_scores, dtype=np.float64)
        if local_status is not None:
            host[per:].view(np.int32)[:mine] = np.asarray(local_status, dtype=np.int32)
        send = torch.from_numpy(host).to(dev)
    recv = torch.empty(world * width, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    allv = recv.cpu().numpy().reshape(world, width)
    out = np.empty(n_total, dtype=np.float64)
    status = np.zeros(n_total, dtype=np.int32)
    for r in range(world):
        k = len(shards[r])
        out[shards[r]] = allv[r, :k]
        status[shards[r]] = allv[r, per:].view(np.int32)[:k]
    return (out, status) if return_status else out


def shard_layout(n_taxa, world_size, trivial=False, size=None):
    """How sp_score_all_splits_shard deals all_splits(taxa) to `world_size` ranks: inside every size class rank r takes
    the combinations r, r + P, ...  Returns (index arrays per rank into the all_splits order, total number of splits);
    rank r's results come in exactly the order of its index array.  Pure arithmetic (no device)."""
    from math import comb

    n = int(n_taxa)
    sizes = [size] if size is not None else list(range(1 if trivial else 2, n // 2 + 1))
    per_rank = [[] for _ in range(world_size)]
    start = 0
    for bal in sizes:
        cnt = comb(n - 1, bal - 1) if 2 * bal == n else comb(n, bal)
        for r in range(world_size):
            per_rank[r].append(np.arange(start + r, start + cnt, world_size, dtype=np.int64))
        start += cnt
    return [np.concatenate(x) if x else np.zeros(0, dtype=np.int64) for x in per_rank], start


def score_all_splits(pattern_probabilities, method=Method.flattening, route="auto", trivial=False, size=None,
                     return_status=False, distributed=None, group=None):
    """Scores of every split of the table's taxa, in `all_splits` order, without building the Python split objects.

    distributed=None: use torch.distributed if it is initialised with world_size > 1 (every rank must call this; every
    rank receives all scores).  Each rank then enumerates and scores ITS shard on the device - the combinations rank,
    rank + P, ... of every size class, an equal share of every cost class -, the kernels' scores and status
    go straight into the send buffer of ONE all_gather, and the host un-permutes (`shard_layout`).  No split list exists
    anywhere: this is BASELINE config 4's partition (20 taxa, 524 267 splits over 8 GPUs, 4 MiB of scores)."""
    al = as_device_alignment(pattern_probabilities)
    code = _method_code(method, route)
    use_dist = distributed
    if use_dist is None:
        try:
            import torch.distributed as dist

            use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        except Exception:
            use_dist = False
    if use_dist:
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        shards, total = shard_layout(al.n_taxa, world, trivial=trivial, size=size)
        per = max(len(s) for s in shards)
        nccl = dist.get_backend(group) == "nccl"
        if nccl:
            send = _send_buffer(packed_width(per))
            score_all_splits_shard(al, code, trivial, size, rank, world, scores_dev_ptr=send.data_ptr(),
                                   status_dev_ptr=send.data_ptr() + per * 8)
            al.ctx.synchronize()      # the collective runs on torch's stream, the kernels on the context's (see _send_buffer)
            out, status = gather_scores(None, shards, total, group=group, device_tensor=send, return_status=True)
        else:
            loc, loc_st = score_all_splits_shard(al, code, trivial, size, rank, world)
            out, status = gather_scores(loc, shards, total, group=group, local_status=loc_st, return_status=True)
        warn_unconverged(status)
        return (out, status) if return_status else out
    if al.n_taxa <= 31:
        # the splits are enumerated on the device (sp_score_all_splits): no split list is built or uploaded at all; on the
        # default flattening route the split descriptors and the launch order are planned on the device as well
        scores, status = score_all_splits_shard(al, code, trivial, size, 0, 1)
        warn_unconverged(status)
        return (scores, status) if return_status else scores
    taxa_arr, a_arr = encode_all_splits(al.n_taxa, trivial=trivial, size=size)
    scores, status = score_encoded(al, taxa_arr, a_arr, code)
    return (scores, status) if return_status else scores


def score_all_splits_shard(al, method_code, trivial, size, rank, world, scores_dev_ptr=None, status_dev_ptr=None):
    """sp_score_all_splits_shard: this rank's share of all_splits, enumerated and scored on the device.  With device
    pointers: enqueue only (results land there); else returns (scores, status) host arrays."""
    lib = al.ctx._lib
    n = C.c_int64()
    args = (al.handle, method_code, int(bool(trivial)), int(size or 0), int(rank), int(world))
    if scores_dev_ptr is not None:
        _lib.check(lib.sp_score_all_splits_shard(*args, C.byref(n), None, C.c_void_p(scores_dev_ptr), None,
                                                 C.c_void_p(status_dev_ptr) if status_dev_ptr else None))
        return n.value
    _lib.check(lib.sp_score_all_splits_shard(*args, C.byref(n), None, None, None, None))
    scores = np.empty(n.value, dtype=np.float64)
    status = np.zeros(n.value, dtype=np.int32)
    if n.value:
        _lib.check(lib.sp_score_all_splits_shard(*args, C.byref(n), _lib._ptr(scores, C.c_double), None,
                                                 _lib._ptr(status, C.c_int32), None), allow_noconv=True)
    return scores, status


def score_splits(pattern_probabilities, splits, method=Method.flattening, distributed=None, group=None,
                 return_status=False, route="auto"):
    """Scores of `splits` (any iterable of the reference's split forms) for one alignment.

    route (flattening method only): "auto" - the sparse in-LDS kernel for count tables, with splits it
    hands back re-scored on the dense route; "dense" - scatter + MFMA Gram + top-4 eigen through HBM;
    "sparse" - sparse kernel only (raises if a split does not fit).

    distributed=None: use torch.distributed if it is initialised with world_size > 1.
    Every rank must call this with the same splits; every rank receives all scores."""
    splits = list(splits)
    al = as_device_alignment(pattern_probabilities)
    code = _method_code(method, route)
    taxa_arr, a_arr = encode_splits(splits, pattern_probabilities, al.n_taxa)
    use_dist = distributed
    if use_dist is None:
        try:
            import torch.distributed as dist

            use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        except Exception:
            use_dist = False
    if not use_dist:
        scores, status = score_encoded(al, taxa_arr, a_arr, code)
        return (scores, status) if return_status else scores
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = shard_indices(split_costs(a_arr, al.n_taxa, code), world)
    mine = shards[rank]
    per = max(len(s) for s in shards)
    if dist.get_backend(group) == "nccl":
        # scores and status of this rank's shard are written by the kernels straight into the packed exchange buffer
        send = _send_buffer(packed_width(per))
        if len(mine):
            score_encoded_async(al, taxa_arr[mine], a_arr[mine], code, send.data_ptr(), send.data_ptr() + per * 8)
        al.ctx.synchronize()          # the collective runs on torch's stream, the kernels on the context's (see _send_buffer)
        out, status = gather_scores(None, shards, len(splits), group=group, device_tensor=send, return_status=True)
        if code in (_lib.SP_METHOD_FLATTENING, _lib.SP_METHOD_FLATTENING_DENSE, _lib.SP_METHOD_FLATTENING_SPARSE) and np.any(status & 3):
            # the asynchronous entry leaves splits without a certificate flagged: every rank finishes them on its own copy
            # of the gathered results (the direct solver is deterministic; none on tree-like tables)
            finish_async(al, taxa_arr, a_arr, out, status)
    else:
        if len(mine):
            loc, loc_st = score_encoded(al, taxa_arr[mine], a_arr[mine], code)
        else:
            loc, loc_st = np.zeros(0), np.zeros(0, dtype=np.int32)
        out, status = gather_scores(loc, shards, len(splits), group=group, local_status=loc_st, return_status=True)
    warn_unconverged(status)
    return (out, status) if return_status else out


class NodeScorer:
    """One process, all GPUs of the node: the library's own multi-GPU partition (sp_node_*, csrc/node.hip) - the candidate
    splits of an alignment sharded over `n_devices` GPUs (index mod P within every size class), one RCCL
    all-gather of the scores over xGMI.  The counterpart for Python programs that run one process per GPU is
    score_all_splits(distributed=True) on torch.distributed; this class drives every GPU from the calling process."""

    def __init__(self, n_devices=0):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.handle = C.c_void_p()
        _lib.check(self._lib.sp_node_create(int(n_devices), C.byref(self.handle)))
        n = C.c_int()
        _lib.check(self._lib.sp_node_info(self.handle, C.byref(n)))
        self.n_devices = int(n.value)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self._lib.sp_node_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def score_all_splits(self, keys, n_taxa, counts=None, weights=None, n_sites=None, method=Method.flattening, route="auto",
                         trivial=False, size=None, return_status=False):
        """Scores of every split of the table's taxa in all_splits order.  keys: packed pattern keys (uint64, taxon 0 most
        significant); counts (with n_sites) or weights: the table's values."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.int64)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if cnt is not None and n_sites is None:
            n_sites = int(cnt.sum())
        code = _method_code(method, route)
        n = C.c_int64()
        args = (self.handle, _lib._ptr(keys, C.c_uint64), _lib._ptr(w, C.c_double), _lib._ptr(cnt, C.c_int64), len(keys),
                int(n_taxa), int(n_sites or 0), code, 1 if trivial else 0, int(size or 0))
        _lib.check(self._lib.sp_node_score_all_splits(*args, C.byref(n), None, None))
        scores = np.zeros(n.value, dtype=np.float64)
        status = np.zeros(n.value, dtype=np.int32)
        _lib.check(self._lib.sp_node_score_all_splits(*args, C.byref(n), _lib._ptr(scores, C.c_double),
                                                      _lib._ptr(status, C.c_int32)), allow_noconv=True)
        warn_unconverged(status)
        return (scores, status) if return_status else scores
