"""Static multi-GPU sharding and the host-side top-r merge.

Every (query, database sequence) score is independent (CPUsearch.c:540-548), so the chunk list of
assemble_multiple_chunks_db is dealt to the ranks once, longest chunk first onto the least-loaded
rank (cost = padded bytes = DP cells per query row); queries and the matrix are replicated.  No
data-path collective: each rank returns its local top-r per query and the lists are merged with the
reference's comparator (score descending, then larger sorted index first).  The reference's own
multi-device scheme is dynamic chunk pulling by one host thread per device (MICsearch.c:53,74-75).
"""
from __future__ import annotations

import numpy as np

from . import host


def assign_chunks(chunk_sizes, world: int):
    """-> owner[i] in [0, world) for every chunk; deterministic LPT (same rule as main.c / swimm_hip_search_chunks)."""
    sizes = np.asarray(chunk_sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(sizes), dtype=np.int64)
    for i in order:
        g = int(np.argmin(load))      # first least-loaded rank
        owner[i] = g
        load[g] += sizes[i]
    return owner


def search_topr_local(engine, q, chunks, owner, rank, sm, open_gap, extend_gap, r, n_valid, device=0, vl=128):
    """top-r of the chunks this rank owns.  engine: 'hip' (MI355X, raises without a GPU) or
    'cpu' (the explicit mode-0 host search; used by -m 0 and by the CPU-only distributed tests)."""
    nq = len(q["m"])
    mine = [c for i, c in enumerate(chunks) if owner[i] == rank]
    if not mine:
        return np.full((nq, r), -1, np.int32), np.full((nq, r), -1, np.int64)
    if engine == "hip":
        from . import hip_backend
        with hip_backend.HipSearcher(device) as s:
            s.set_queries(q["a"], q["m"], q["disp"], sm, open_gap, extend_gap)
            for c in mine:
                s.add_chunk(c["b"], c["n"], c["disp"], vl, c["first_group"])
            ts, ti, _ = s.search_topr(r, n_valid)
        return ts, ti
    if engine != "cpu":
        raise ValueError(engine)
    lists_s, lists_i = [], []
    for c in mine:
        disp = np.concatenate([c["disp"].astype(np.uint64), [np.uint64(c["vD"])]])
        sc, _ = host.cpu_search(q["a"], q["m"], q["disp"], c["b"], c["n"], disp, sm, open_gap, extend_gap, vl, threads=2)
        first = c["first_group"] * vl
        keep = max(0, min(sc.shape[1], n_valid - first))
        cs, ci = [], []
        for qi in range(nq):
            s_, i_ = host.topr(sc[qi, :keep], r)
            cs.append(s_); ci.append(np.where(i_ >= 0, i_ + first, -1))
        lists_s.append(np.stack(cs)); lists_i.append(np.stack(ci))
    ts = np.zeros((nq, r), np.int32); ti = np.zeros((nq, r), np.int64)
    for qi in range(nq):
        ts[qi], ti[qi] = host.topr_merge(np.stack([l[qi] for l in lists_s]), np.stack([l[qi] for l in lists_i]), r)
    return ts, ti


def allgather_merge(ts, ti, dist=None, group=None, index_base=0):
    """The path's only exchange step: every rank's top-r per query -> the merged listing on every rank (host merge, utils.c
    order).  `dist`: torch.distributed (any backend) or None for one rank; `group`: the process group that carries the lists
    (bench.py: the RCCL group; default: the default group); tensors live where that group's backend needs them.
    `index_base` is added to this rank's indices first (ranks that hold disjoint databases number their sequences from 0)."""
    nq, r = ts.shape
    if dist is None or dist.get_world_size() == 1:
        return ts, ti
    import torch
    mine = torch.from_numpy(np.concatenate([ts.astype(np.int64).ravel(), np.where(ti >= 0, ti + index_base, -1).astype(np.int64).ravel()]))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = mine.to(dev)
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine, group=group)
    g = torch.stack(allv).cpu().numpy()
    ms = np.zeros((nq, r), np.int32); mi = np.zeros((nq, r), np.int64)
    for qi in range(nq):
        ls = g[:, qi * r:(qi + 1) * r].astype(np.int32)
        li = g[:, nq * r + qi * r: nq * r + (qi + 1) * r]
        ms[qi], mi[qi] = host.topr_merge(ls, li, r)
    return ms, mi
