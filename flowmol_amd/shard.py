"""Multi-GPU sampling: molecules are independent, so a batch shards embarrassingly across the ranks of
one node (one process per GPU, ``torch.distributed`` with backend "nccl" = RCCL over xGMI) with no
collective during integration and ONE all-gather of the packed results at the end.

The reference has no multi-GPU sampling path (test.py:91 hard-codes cuda:0); this is the MI355X-native
addition named by BASELINE.json's north star.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def partition_lpt(n_atoms: torch.Tensor, world_size: int) -> List[torch.Tensor]:
    """Greedy longest-processing-time assignment balancing the per-rank cost sum n_i*(n_i-1)
    (work is proportional to directed edges).  Returns, per rank, the ORIGINAL indices it owns
    (ascending, so the within-rank order is the caller's order).  Deterministic."""
    n = n_atoms.to(torch.int64).cpu()
    cost = (n * (n - 1)).tolist()
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0] * world_size
    owner = [0] * len(cost)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cost[i]
    return [torch.tensor([i for i in range(len(cost)) if owner[i] == r], dtype=torch.int64) for r in range(world_size)]


def pack_results(x: torch.Tensor, a: torch.Tensor, c: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """Pack one rank's results into a flat uint8 buffer: x fp32 (N,3) | a u8 (N) | c u8 (N) | e u8 (U)."""
    return torch.cat([x.contiguous().view(torch.uint8).reshape(-1), a.to(torch.uint8), c.to(torch.uint8), e.to(torch.uint8)])


def unpack_results(buf: torch.Tensor, N: int, U: int) -> Dict[str, torch.Tensor]:
    o = 0
    x = buf[o:o + N * 12].contiguous().view(torch.float32).reshape(N, 3); o += N * 12
    a = buf[o:o + N].to(torch.int32); o += N
    c = buf[o:o + N].to(torch.int32); o += N
    e = buf[o:o + U].to(torch.int32)
    return {'x': x, 'a': a, 'c': c, 'e': e}


def gather_results(local: Dict[str, torch.Tensor], n_atoms_all: torch.Tensor, parts: List[torch.Tensor], group=None
                   ) -> Dict[str, torch.Tensor]:
    """One all_gather of the padded packed buffers; every rank returns the full batch in the ORIGINAL
    molecule order.  ``local`` holds this rank's x (N_r,3), a, c (N_r) and e (U_r) in its own order."""
    world = dist.get_world_size(group)
    n_all = n_atoms_all.to(torch.int64).cpu()
    sizes = []
    for r in range(world):
        nr = n_all[parts[r]]
        sizes.append((int(nr.sum()), int((nr * (nr - 1) // 2).sum())))
    nbytes = [N * 14 + U for N, U in sizes]
    cap = (max(nbytes) + 15) // 16 * 16          # 16-byte slots: the fp32 view of a rank's slice needs 4-byte alignment
    dev = local['x'].device
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine = pack_results(local['x'], local['a'], local['c'], local['e'])
    send[:mine.numel()] = mine
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    # scatter back to the original molecule order (index arithmetic vectorised: no per-molecule Python work)
    Ntot = int(n_all.sum())
    pairs_all = n_all * (n_all - 1) // 2
    Utot = int(pairs_all.sum())
    node_off = torch.cumsum(n_all, 0) - n_all
    pair_off = torch.cumsum(pairs_all, 0) - pairs_all
    out = {'x': torch.empty(Ntot, 3, device=dev), 'a': torch.empty(Ntot, dtype=torch.int32, device=dev),
           'c': torch.empty(Ntot, dtype=torch.int32, device=dev), 'e': torch.empty(Utot, dtype=torch.int32, device=dev)}
    for r in range(world):
        N, U = sizes[r]
        got = unpack_results(recv[r * cap:r * cap + nbytes[r]], N, U)
        nidx = _ranges(node_off[parts[r]].to(dev), n_all[parts[r]].to(dev))      # index arithmetic on the device: only the
        pidx = _ranges(pair_off[parts[r]].to(dev), pairs_all[parts[r]].to(dev))  # per-molecule offsets cross PCIe
        out['x'][nidx] = got['x']
        out['a'][nidx] = got['a']
        out['c'][nidx] = got['c']
        out['e'][pidx] = got['e']
    return out


def _ranges(starts: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """Concatenation of arange(starts[i], starts[i] + lens[i]) for all i."""
    total = int(lens.sum())
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=starts.device)
    first = torch.cumsum(lens, 0) - lens                      # position of each range in the output
    return torch.arange(total, dtype=torch.int64, device=starts.device) + torch.repeat_interleave(starts - first, lens, output_size=total)
