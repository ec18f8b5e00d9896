"""CPU, world_size 2, gloo: the N > 1 path (shard plan -> per-rank scoring -> all_gather ->
un-permute) with a stand-in scorer (no GPU here; the stand-in is the oracle, used purely as
the checker's arithmetic - the product's scorer is the HIP library)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import splitp_oracle as O
    from splitp_amd import _lib, batch
    from splitp_amd import synthetic as syn

    n = 6
    names = syn.taxa_names(n)
    sites = syn.simulate_sites(n, 2000, 0.05, seed=5)
    keys, counts = syn.pattern_table(sites)
    splits = list(O.all_splits(names))

    class T(dict):
        pass

    t = T()
    t.taxa = tuple(names)
    taxa_arr, a_arr = batch.encode_splits(splits, t, n)
    shards = batch.shard_indices(batch.split_costs(a_arr, n, _lib.SP_METHOD_FLATTENING), world)
    mine = shards[rank]
    local = np.array([O.score_from_matrix_gram(O.reduced_flattening_packed(
        keys, counts, n, taxa_arr[i, : a_arr[i]], taxa_arr[i, a_arr[i]:])[0].astype(float)) for i in mine])
    allv = batch.gather_scores(local, shards, len(splits))
    full = np.array([O.score_from_matrix_gram(O.reduced_flattening_packed(
        keys, counts, n, taxa_arr[i, : a_arr[i]], taxa_arr[i, a_arr[i]:])[0].astype(float)) for i in range(len(splits))])
    ok = np.array_equal(allv, full)
    if rank == 0:
        np.save(out_path, np.array([ok, len(splits), len(mine)], dtype=np.int64))
    # every rank must hold the same, complete vector
    tchk = torch.tensor([float(ok)])
    dist.all_reduce(tchk, op=dist.ReduceOp.MIN)
    assert tchk.item() == 1.0
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_scores_all_gather_gloo(tmp_path):
    out = str(tmp_path / "res.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    assert res[0] == 1 and res[1] == 2 ** 5 - 6 - 1


def _worker_score_splits(rank, world, port, out_path):
    """The real batch.score_splits(distributed=True) code path - shard plan, per-rank scoring call, ONE all_gather of
    the packed scores + status, un-permute, warning - with only the device scorer stubbed (no GPU here)."""
    import warnings

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import splitp_oracle as O
    from splitp_amd import batch
    from splitp_amd import synthetic as syn

    n = 6
    names = syn.taxa_names(n)
    sites = syn.simulate_sites(n, 2000, 0.05, seed=5)
    keys, counts = syn.pattern_table(sites)
    splits = list(O.all_splits(names))

    class FakeAlignment:
        n_taxa = n
        taxa = tuple(names)

    calls = []

    def fake_score_encoded(al, split_taxa, split_a, method_code, scores_dev_ptr=None, want_host=True):
        calls.append(len(split_a))
        sc = np.array([O.score_from_matrix_gram(O.reduced_flattening_packed(
            keys, counts, n, split_taxa[i, : split_a[i]], split_taxa[i, split_a[i]:])[0].astype(float))
            for i in range(len(split_a))])
        st = (np.arange(len(split_a), dtype=np.int32) + 7) << 8           # "operator applications" per split
        if rank == 1 and len(st):
            st[0] |= 1                                                     # one unconverged split on rank 1
        return sc, st

    batch.score_encoded = fake_score_encoded
    batch.as_device_alignment = lambda table, device=None: FakeAlignment()
    fake = FakeAlignment()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        scores, status = batch.score_splits(fake, splits, distributed=True, return_status=True)
        only_scores = batch.score_splits(fake, splits, distributed=True)
    warned = sum(issubclass(w.category, RuntimeWarning) for w in caught)
    full = np.array([O.score_from_matrix_gram(O.reduced_flattening_packed(
        keys, counts, n, list(map(names.index, s[0])), list(map(names.index, s[1])))[0].astype(float)) for s in splits])
    ok = (np.array_equal(scores, full) and np.array_equal(only_scores, full) and status.dtype == np.int32
          and int(np.count_nonzero(status & 1)) == 1 and np.all(status >> 8 >= 7) and warned == 2
          and len(calls) == 2 and calls[0] < len(splits))
    tchk = torch.tensor([float(ok)])
    dist.all_reduce(tchk, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.save(out_path, np.array([int(tchk.item()), len(splits), calls[0]], dtype=np.int64))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_score_splits_distributed_path_gloo(tmp_path):
    out = str(tmp_path / "res2.npy")
    port = _free_port()
    mp.spawn(_worker_score_splits, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    assert res[0] == 1 and res[1] == 2 ** 5 - 6 - 1 and 0 < res[2] < res[1]


def _worker_all_splits(rank, world, port, out_path):
    """batch.score_all_splits(distributed=True): shard_layout + ONE all_gather + un-permute, device call stubbed."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splitp_amd import batch

    n = 9

    class FakeAlignment:
        n_taxa = n

    ta, aa = batch.encode_all_splits(n)
    shards, total = batch.shard_layout(n, world)
    truth = 0.001 * np.arange(total) + 1.0            # the "score" of split i in all_splits order

    def fake_shard(al, method_code, trivial, size, r, w, scores_dev_ptr=None, status_dev_ptr=None):
        assert (r, w) == (rank, world)
        idx, _ = batch.shard_layout(n, w, trivial=trivial, size=size)
        return truth[idx[r]].copy(), ((idx[r] % 5).astype(np.int32) << 8)

    batch.score_all_splits_shard = fake_shard
    batch.as_device_alignment = lambda table, device=None: FakeAlignment()
    got, st = batch.score_all_splits(FakeAlignment(), distributed=True, return_status=True)
    ok = total == len(aa) and np.array_equal(got, truth) and np.array_equal(st >> 8, np.arange(total) % 5)
    tchk = torch.tensor([float(ok)])
    dist.all_reduce(tchk, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.save(out_path, np.array([int(tchk.item()), total, len(shards[0])], dtype=np.int64))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_score_all_splits_distributed_gloo(tmp_path):
    out = str(tmp_path / "res3.npy")
    port = _free_port()
    mp.spawn(_worker_all_splits, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    assert res[0] == 1 and res[1] == 2 ** 8 - 9 - 1 and abs(2 * res[2] - res[1]) <= 4


def test_shard_layout_is_a_partition_in_all_splits_order():
    sys.path.insert(0, ROOT)
    from splitp_amd import batch

    for n in (4, 5, 8, 10, 11):
        for kw in ({}, {"trivial": True}, {"size": 2}, {"size": n // 2}):
            _, aa = batch.encode_all_splits(n, **kw)
            for world in (1, 2, 3, 8):
                shards, total = batch.shard_layout(n, world, **kw)
                assert total == len(aa)
                assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(total))
                k = np.minimum(aa, n - aa)
                for kk in np.unique(k):            # every rank gets an equal share (+-1) of every size class
                    share = [int(np.count_nonzero(k[s] == kk)) for s in shards]
                    assert max(share) - min(share) <= 1, (n, kw, world, kk, share)
