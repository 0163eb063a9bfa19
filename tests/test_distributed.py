"""world_size-2 gloo test of the multi-GPU path's host logic: molecule sharding (LPT) and the single
all-gather of packed results restoring the original molecule order.  (The data path has no other collective.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowmol_amd.shard import gather_results, partition_lpt


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_atoms, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    parts = partition_lpt(n_atoms, world)
    mine = parts[rank]
    # fabricate per-molecule results that encode the ORIGINAL molecule index
    xs, as_, cs, es = [], [], [], []
    for i in mine.tolist():
        n = int(n_atoms[i])
        u = n * (n - 1) // 2
        xs.append(torch.full((n, 3), float(i)) + torch.arange(n).float()[:, None] * 0.01)
        as_.append(torch.full((n,), i % 11, dtype=torch.int32))
        cs.append(torch.full((n,), i % 6, dtype=torch.int32))
        es.append(torch.full((u,), i % 5, dtype=torch.int32))
    local = {'x': torch.cat(xs), 'a': torch.cat(as_), 'c': torch.cat(cs), 'e': torch.cat(es)}
    full = gather_results(local, n_atoms, parts)
    ok = True
    noff = poff = 0
    for i, n in enumerate(n_atoms.tolist()):
        u = n * (n - 1) // 2
        ok &= bool(torch.allclose(full['x'][noff:noff + n, 0], torch.full((n,), float(i)) + torch.arange(n).float() * 0.01))
        ok &= bool((full['a'][noff:noff + n] == i % 11).all()) and bool((full['e'][poff:poff + u] == i % 5).all())
        noff += n
        poff += u
    q.put((rank, ok, int(full['x'].shape[0])))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('sizes', [[5, 47, 12, 30, 8, 64, 3, 21, 47],
                                   [3, 2]])      # packed payloads of 45 and 29 bytes: slot size must be padded for the fp32 view
def test_shard_and_gather_world2(sizes):
    n_atoms = torch.tensor(sizes)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_atoms, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(n == int(n_atoms.sum()) for _, _, n in res)
