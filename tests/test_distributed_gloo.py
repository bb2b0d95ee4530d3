"""N > 1 path on CPU: world_size 2 over gloo.  The chunk list of the golden database is sharded
statically, each rank searches only its chunks (explicit host-CPU engine here; the MI355X engine is
the same call with engine='hip'), the per-rank top-r lists are all-gathered and merged; the result
must equal the reference's global sorted listing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, load_npy


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, prefix, qfa, r, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from swimm_amd import host, sharding, submat
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = host.db_load(prefix)
    q = host.queries_load(qfa, False)
    ch = host.Chunks(db["lengths"], db["codes"], 128, 20000)
    owner = sharding.assign_chunks([c["vD"] for c in ch.chunks], world)
    assert set(owner.tolist()) == set(range(world))          # every rank got work
    ts, ti = sharding.search_topr_local("cpu", q, ch.chunks, owner, rank, submat.table("blosum62"), 10, 2, r, db["count"])
    # (bench.py hands over the process group that carries the lists -- its RCCL group; here a gloo subgroup stands in for it)
    grp = dist.new_group(ranks=list(range(world)), backend="gloo") if r == 100 else None
    ms, mi = sharding.allgather_merge(ts, ti, dist, group=grp)
    np.save(os.path.join(outdir, f"s{rank}.npy"), ms); np.save(os.path.join(outdir, f"i{rank}.npy"), mi)
    np.save(os.path.join(outdir, f"own{rank}.npy"), owner)
    ch.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("r", [10, 100])
def test_two_rank_shard_and_merge(tmp_path, golden, r):
    from swimm_amd import host
    prefix = str(tmp_path / "db")
    host.preprocess_db(os.path.join(GOLDEN, golden["db_fasta"]), prefix)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, prefix, os.path.join(GOLDEN, golden["query_fasta"]), r, str(tmp_path)), nprocs=2, join=True)
    sc, order = load_npy("scores_blosum62_g10_e2.npy"), load_npy("order_blosum62_g10_e2.npy")
    for rank in range(2):
        ms, mi = np.load(tmp_path / f"s{rank}.npy"), np.load(tmp_path / f"i{rank}.npy")
        for qi in range(sc.shape[0]):
            assert np.array_equal(mi[qi], order[qi][:r]) and np.array_equal(ms[qi], sc[qi][order[qi][:r]])
    assert np.array_equal(np.load(tmp_path / "own0.npy"), np.load(tmp_path / "own1.npy"))   # same static plan on every rank


def test_assign_chunks_is_balanced_lpt():
    from swimm_amd import sharding
    sizes = [100, 90, 80, 10, 10, 10, 5]
    owner = sharding.assign_chunks(sizes, 3)
    loads = [sum(s for s, o in zip(sizes, owner) if o == g) for g in range(3)]
    assert sorted(loads) == [100, 100, 105] and owner.tolist()[:3] == [0, 1, 2]
    assert sharding.assign_chunks([7], 4).tolist() == [0]


def _worker_disjoint(rank, world, port, outdir):
    """ranks that hold DISJOINT databases (bench.py's weak-scaling c2: every rank numbers its sequences from 0)"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from swimm_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ts = np.array([[50 - rank, 40, 30 + rank], [7, 7, -1]], np.int32)      # second query: a short list (one empty slot)
    ti = np.array([[5, 3, 1], [9, 2, -1]], np.int64)
    ms, mi = sharding.allgather_merge(ts, ti, dist, index_base=rank * 1000)
    np.save(os.path.join(outdir, f"ds{rank}.npy"), ms); np.save(os.path.join(outdir, f"di{rank}.npy"), mi)
    dist.barrier()
    dist.destroy_process_group()


def test_merge_of_disjoint_databases_offsets_the_indices(tmp_path):
    mp.spawn(_worker_disjoint, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        ms, mi = np.load(tmp_path / f"ds{rank}.npy"), np.load(tmp_path / f"di{rank}.npy")
        assert ms.tolist() == [[50, 49, 40], [7, 7, 7]]
        assert mi.tolist() == [[5, 1005, 1003], [1009, 1002, 9]]            # ties: the larger index first (utils.c:12,52)
