"""world_size-2 gloo tests of the sharding logic (no GPU): image shards, pair shards and the
descriptor all-gather reassemble exactly the single-process problem."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import matcher_oracle as mo
from util_data import image_set
from vit_colmap_amd import dist as vd


def test_shard_and_pair_partition_cover_everything():
    for n, world in [(50, 1), (50, 2), (200, 8), (7, 4), (3, 8)]:
        blocks = [vd.shard_range(n, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        allp = np.concatenate([vd.pairs_for_rank(n, r, world) for r in range(world)])
        ref = mo.exhaustive_pairs(n)
        assert len(allp) == len(ref)
        assert {tuple(p) for p in allp} == {tuple(p) for p in ref}
        sizes = [len(vd.pairs_for_rank(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
        assert np.array_equal(vd.pair_index(n, ref[:, 0], ref[:, 1]), np.arange(len(ref)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_max, d = 40, 64
        desc, counts = image_set(21, n_images, n_max, d, kind="scene", noise=0.1)
        per = (n_images + world - 1) // world
        lo, hi = vd.shard_range(n_images, rank, world)
        local = np.zeros((per, n_max, d), np.uint8)
        lc = np.zeros(per, np.int32)
        local[: hi - lo] = desc[lo:hi]
        lc[: hi - lo] = counts[lo:hi]
        all_desc, all_counts = vd.all_gather_descriptors(torch.from_numpy(local), torch.from_numpy(lc))
        all_desc, all_counts = all_desc.numpy()[:n_images], all_counts.numpy()[:n_images]
        ok = np.array_equal(all_desc, desc) and np.array_equal(all_counts, counts)
        # each rank matches its pair shard (oracle stands in for the GPU kernel in this CPU test)
        pairs = vd.pairs_for_rank(n_images, rank, world)
        lists = [mo.match_pair(all_desc[a, : all_counts[a]], all_desc[b, : all_counts[b]]) for a, b in pairs]
        cnt = np.array([len(x) for x in lists], np.int32)
        m = np.zeros((len(pairs), n_max, 2), np.uint32)
        for p, x in enumerate(lists):
            m[p, : len(x)] = x
        gathered = vd.gather_match_lists(pairs, cnt, m, dst=0)
        if rank == 0:
            got = {}
            for prs, c, mm in gathered:
                for p, (a, b) in enumerate(prs):
                    got[(int(a), int(b))] = mm[p, : c[p]]
            ref_pairs = mo.exhaustive_pairs(n_images)
            ok = ok and len(got) == len(ref_pairs)
            for a, b in ref_pairs:
                ok = ok and np.array_equal(got[(int(a), int(b))],
                                           mo.match_pair(desc[a, : counts[a]], desc[b, : counts[b]]))
            q.put(bool(ok))
        else:
            assert gathered is None
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [6, 5])
def test_two_rank_gloo_pipeline(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)
