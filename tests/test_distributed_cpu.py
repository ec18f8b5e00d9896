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
