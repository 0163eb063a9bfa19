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


FRAME_KEYS = ('x', 'x_1_pred', 'a', 'c', 'e', 'a_1_pred', 'c_1_pred', 'e_1_pred')      # packing order: the fp32 blocks first (4-byte aligned views)


def slice_prior(prior: dict, n_atoms_all: torch.Tensor, mine: torch.Tensor) -> dict:
    """The rows of a reference-format prior dict of the WHOLE batch (flowmol.py:534-545: x_0 (N,3), a_0 / c_0 (N,*), e_0 per directed edge (E,*) with
    every molecule's upper block first, or per unordered pair (U,*)) that belong to the molecules ``mine`` (original indices, ascending)."""
    n = n_atoms_all.to(torch.int64).cpu()
    pairs = n * (n - 1) // 2
    nidx = _ranges((torch.cumsum(n, 0) - n)[mine], n[mine])
    out = dict(prior)
    for k in ('x_0', 'a_0', 'c_0'):
        out[k] = prior[k][nidx.to(prior[k].device)]
    e0 = prior['e_0']
    if e0.shape[0] == int(2 * pairs.sum()) and int(pairs.sum()) > 0:
        eidx = _ranges((torch.cumsum(2 * pairs, 0) - 2 * pairs)[mine], 2 * pairs[mine])
    else:
        eidx = _ranges((torch.cumsum(pairs, 0) - pairs)[mine], pairs[mine])
    out['e_0'] = e0[eidx.to(e0.device)]
    return out


def gather_frames(local, n_atoms_all: torch.Tensor, parts: List[torch.Tensor], n_frames: int, device, group=None) -> Dict[str, torch.Tensor]:
    """The trajectory frames of a sharded run (``sample_distributed(xt_traj / ep_traj)``): ONE more all_gather, of every rank's packed frames --
    x / x_1_pred as fp32, tokens as bytes: 24 B per atom and 2 B per pair and frame -- returning the full batch's frames in the ORIGINAL molecule
    order on every rank.  ``local`` = this rank's frames on the device ('x', 'a', 'c', 'e': n_frames; '*_1_pred': n_frames - 1), None for a rank
    that owns no molecule."""
    world = dist.get_world_size(group)
    n_all = n_atoms_all.to(torch.int64).cpu()
    pairs_all = n_all * (n_all - 1) // 2
    T = {k: (n_frames - 1 if k.endswith('_1_pred') else n_frames) for k in FRAME_KEYS}
    sizes = [(int(n_all[parts[r]].sum()), int(pairs_all[parts[r]].sum())) for r in range(world)]

    def nbytes(N, U):
        return sum(T[k] * ((N * 12) if k.startswith('x') else (U if k.startswith('e') else N)) for k in FRAME_KEYS)
    cap = (max(nbytes(N, U) for N, U in sizes) + 15) // 16 * 16
    send = torch.zeros(cap, dtype=torch.uint8, device=device)
    if local is not None:
        mine = torch.cat([(local[k].contiguous().view(torch.uint8) if k.startswith('x') else local[k].to(torch.uint8)).reshape(-1) for k in FRAME_KEYS])
        send[:mine.numel()] = mine
    recv = torch.empty(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    Ntot, Utot = int(n_all.sum()), int(pairs_all.sum())
    node_off, pair_off = torch.cumsum(n_all, 0) - n_all, torch.cumsum(pairs_all, 0) - pairs_all
    out = {k: (torch.empty(T[k], Ntot, 3, device=device) if k.startswith('x') else torch.empty(T[k], Utot if k.startswith('e') else Ntot, dtype=torch.int32, device=device))
           for k in FRAME_KEYS}
    for r in range(world):
        N, U = sizes[r]
        if N == 0:
            continue
        nidx = _ranges(node_off[parts[r]].to(device), n_all[parts[r]].to(device))
        pidx = _ranges(pair_off[parts[r]].to(device), pairs_all[parts[r]].to(device))
        o = r * cap
        for k in FRAME_KEYS:
            if k.startswith('x'):
                nb = T[k] * N * 12
                out[k][:, nidx] = recv[o:o + nb].contiguous().view(torch.float32).reshape(T[k], N, 3)
            else:
                rows = U if k.startswith('e') else N
                nb = T[k] * rows
                out[k][:, pidx if k.startswith('e') else nidx] = recv[o:o + nb].reshape(T[k], rows).to(torch.int32)
            o += nb
    return out


def _ranges(starts: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """Concatenation of arange(starts[i], starts[i] + lens[i]) for all i."""
    total = int(lens.sum())
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=starts.device)
    first = torch.cumsum(lens, 0) - lens                      # position of each range in the output
    return torch.arange(total, dtype=torch.int64, device=starts.device) + torch.repeat_interleave(starts - first, lens, output_size=total)
