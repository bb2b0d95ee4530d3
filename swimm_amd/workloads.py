"""The BASELINE.json configurations as in-memory inputs of the search path (synthetic, deterministic).

One builder for bench.py, tools/bench_configs.py and tests/test_gpu_configs.py: a length-sorted
database (lengths + recoded residues, as the .seq file stores them, sequences.c:201-205) with planted
homologs of every query, and the query batch in ascending-length order (sequences.c:344).

  c2  375-aa query (P07327-shaped) x 1M log-normal proteins, BLOSUM62
  c3  20-query set x Swiss-Prot-shaped database (0.01 % tail to 35 000 residues), BLOSUM50
  c4  5478-aa query (Q9UKN1-shaped) x Env-NR-shaped database, BLOSUM62
  c5  20-query set x Env-NR-shaped database (35.5M sequences, 7e9 residues at scale 1), PAM250
"""
from __future__ import annotations

import numpy as np

from . import host, synth

CONFIGS = {
    "c2": dict(lengths="c2", queries=[3], matrix="blosum62", seed=2),
    "c2long": dict(lengths="c2", queries=[19], matrix="blosum62", seed=2),
    "c3": dict(lengths="c3", queries=list(range(20)), matrix="blosum50", seed=3),
    "c4": dict(lengths="c4", queries=[19], matrix="blosum62", seed=5),
    "c5": dict(lengths="c5", queries=list(range(20)), matrix="pam250", seed=5),
    "c3clip": dict(lengths="c3", queries=list(range(20)), matrix="blosum50", seed=3, clip=3000),   # c3 without its long-sequence tail
}


_FULL_N = {"c2": 1_000_000, "c3": 540_000, "c4": 35_500_000, "c5": 35_500_000}


class SortedDb:
    """A configuration's database described by its sorted length vector; residues are generated on demand for any
    range of sorted positions (the generator is counter-based, the background residues i.i.d.), so a rank of a
    sharded run only ever materialises its own slabs.  Planted homologs sit at their sorted positions."""

    def __init__(self, name: str, scale: float = 1.0, queries=None, seed=None, n_sequences=None):
        cfg = dict(CONFIGS[name])
        self.name, self.matrix = name, cfg["matrix"]
        self.seed = cfg["seed"] if seed is None else seed
        qidx = cfg["queries"] if queries is None else [cfg["queries"][i] for i in queries]
        qs_all = synth.make_queries(self.seed)
        qs = [qs_all[i] for i in qidx]
        if n_sequences is not None:
            scale = n_sequences / _FULL_N[cfg["lengths"]]
        self.scale = scale
        base = synth.config_lengths(cfg["lengths"], scale)
        if cfg.get("clip"):
            base = np.minimum(base, cfg["clip"])
        planted = synth.planted_homologs(self.seed, qs)
        lens = np.concatenate([base, np.array([len(s) for _, s in planted], dtype=np.int64)])
        order = np.argsort(lens, kind="stable")
        self.lengths = lens[order].astype(np.uint16)
        self.offs = np.concatenate([[0], np.cumsum(self.lengths.astype(np.int64))])
        self.n, self.residues = len(lens), int(lens.sum())
        pos_of = np.empty(len(lens), dtype=np.int64)
        pos_of[order] = np.arange(len(lens))
        self.planted = sorted((int(pos_of[len(base) + k]), host.recode(seq)) for k, (_, seq) in enumerate(planted))
        qorder = np.argsort([len(s) for _, s in qs], kind="stable")
        qa = [host.recode(qs[i][1]) for i in qorder]
        self.a = np.concatenate(qa)
        self.m = np.array([len(x) for x in qa], dtype=np.uint16)
        self.disp = np.concatenate([[0], np.cumsum(self.m)]).astype(np.uint32)
        self.query_residues = int(self.m.astype(np.int64).sum())

    def codes(self, s0: int, s1: int) -> np.ndarray:
        """recoded residues of the sorted sequences [s0, s1), concatenated"""
        b0, b1 = int(self.offs[s0]), int(self.offs[s1])
        out = np.empty(b1 - b0, dtype=np.int8)
        blk = 1 << 26
        for s in range(b0, b1, blk):
            e = min(b1, s + blk)
            out[s - b0:e - b0] = host.recode(synth.residues(self.seed, 7, s, e - s))
        for p, seq in self.planted:
            if s0 <= p < s1:
                out[self.offs[p] - b0:self.offs[p] - b0 + len(seq)] = seq
        return out

    def slabs(self, parts: int):
        """[(s0, s1, padded_bytes)]: `parts` runs of whole 128-sequence groups with about equal padded size (the cost of
        a slab: its groups' longest members x 128, i.e. the DP cells per query row)"""
        ng = (self.n + 127) // 128
        last = np.minimum((np.arange(ng) + 1) * 128 - 1, self.n - 1)
        gbytes = ((self.lengths[last].astype(np.int64) + 3) // 4 * 4) * 128
        cum = np.concatenate([[0], np.cumsum(gbytes)])
        parts = max(1, min(parts, ng))
        cuts = [0]
        for k in range(1, parts):
            g = int(np.searchsorted(cum, cum[-1] * k / parts))
            cuts.append(min(max(g, cuts[-1] + 1), ng - (parts - k)))
        cuts.append(ng)
        return [(cuts[k] * 128, min(cuts[k + 1] * 128, self.n), int(cum[cuts[k + 1]] - cum[cuts[k]])) for k in range(parts)]


def build(name: str, scale: float, queries=None, seed=None, n_sequences=None):
    """-> dict(lengths uint16 sorted, codes int8, offs int64, a, m, disp, matrix, residues, n).
    `queries`: indices into the config's query list (default: all).  `n_sequences` overrides scale."""
    db = SortedDb(name, scale, queries, seed, n_sequences)
    return {"name": name, "scale": db.scale, "lengths": db.lengths, "codes": db.codes(0, db.n), "offs": db.offs, "a": db.a, "m": db.m,
            "disp": db.disp, "matrix": db.matrix, "residues": db.residues, "n": db.n, "query_residues": db.query_residues}
