"""Randomised parity: databases, query batches, penalties, upload paths and library options drawn from a seed, the whole
score matrix against the CPU checker (the reference's AVX2 path when oracle/_ref is built).  Every seed is a fixed
case; a failure names the seed and the options.  SWIMM_FUZZ_FIRST / SWIMM_FUZZ_SEEDS widen the sweep (default: seeds 0..159), SWIMM_FUZZ_GE_MAX the extend penalties (default 0..5)."""
import os

import numpy as np
import pytest

from helpers import oracle_matrix
from swimm_amd import hip_backend, host, submat

pytestmark = pytest.mark.gpu

MATRICES = ["blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250"]


def draw_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 100, 129, 1500, 6000, 20000, 20000, 70000]))
    mean = float(rng.choice([30, 120, 350]))
    L = np.clip(rng.lognormal(np.log(mean), 0.6, n), 1, 3000).astype(np.int64)
    for _ in range(int(rng.integers(0, 4))):                      # a few long sequences: the lane-systolic tail
        L[rng.integers(0, n)] = int(rng.integers(2000, 9000))
    if rng.random() < 0.3:
        L[rng.integers(0, n)] = 0                                  # an empty record
    L = np.sort(L).astype(np.uint16)
    total = int(L.astype(np.int64).sum())
    codes = rng.integers(0, 23, total).astype(np.int8)
    offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
    nq = int(rng.choice([1, 2, 3, 5, 9, 14]))
    qlens = np.sort(np.clip(rng.lognormal(np.log(float(rng.choice([20, 150, 700]))), 0.8, nq), 1, 2600).astype(np.int64))
    queries = []
    for ql in qlens:
        q = rng.integers(0, 23, int(ql)).astype(np.int8)
        if total and rng.random() < 0.6:                           # plant a piece of a database sequence: real alignments
            i = int(rng.integers(0, n))
            if L[i] > 4:
                k = int(min(L[i], ql, rng.integers(4, 400)))
                q[:k] = codes[offs[i]:offs[i] + k]
        queries.append(q)
    m = np.array([len(q) for q in queries], np.uint16)
    disp = np.concatenate([[0], np.cumsum(m.astype(np.int64))]).astype(np.uint32)
    w = {"lengths": L, "codes": codes, "offs": offs, "n": n, "residues": total, "a": np.concatenate(queries), "m": m, "disp": disp,
         "query_residues": int(m.astype(np.int64).sum()), "matrix": str(rng.choice(MATRICES))}
    go, ge = int(rng.integers(0, 21)), int(rng.integers(0, int(os.environ.get("SWIMM_FUZZ_GE_MAX", "6"))))     # (the binary16 tier's offset period depends on ge)
    opts = {}
    pool = [("resident", [0, 1]), ("dynamic", [0]), ("tail_mode", [1, 2]), ("f16", [0]), ("force_i32", [1]), ("wg_limit", [4, 64]),
            ("bnd_mib", [1]), ("score_mib", [1]), ("tail_frac", [10, 200]),
            ("max_waves", [1, 4, 8]), ("upload_piece_kib", [16, 64, 1024]), ("tail_cap", [0, 5, 80]),
            ("sp_threshold", [0, 64, 500]), ("stack", [0]), ("cut", [0, 5, 200])]
    for key, vals in pool:
        if rng.random() < 0.2:
            opts[key] = int(rng.choice(vals))
    if rng.random() < 0.35 and "force_i32" not in opts and "f16" not in opts:
        T = int(rng.choice([8, 12, 16, 20, 24, 28, 32, 36]))
        opts["rows_per_wave"] = T
        opts["waves"] = int(rng.integers(1, (12 if T > 28 else 16) + 1))
    path = str(rng.choice(["chunks", "chunks_lazy", "slabs", "slabs_lazy"]))
    return w, go, ge, opts, path, rng


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWIMM_FUZZ_FIRST", "0")), int(os.environ.get("SWIMM_FUZZ_FIRST", "0")) + int(os.environ.get("SWIMM_FUZZ_SEEDS", "160"))))
def test_random_case_against_the_checker(seed):
    w, go, ge, opts, path, rng = draw_case(seed)
    sm = submat.table(w["matrix"])
    chunks = None
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_option("lazy_upload", 1 if path.endswith("lazy") else 0)
        s.set_queries(w["a"], w["m"], w["disp"], sm, go, ge)
        if path.startswith("chunks"):
            chunks = host.Chunks(w["lengths"], w["codes"], 128, int(rng.choice([2000, 50000, 1 << 20, 96 << 20])))
            for c in chunks.chunks:
                s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
            stride = chunks.vc * 128
        else:
            cuts = sorted(set([0, w["n"]] + [int(x) // 128 * 128 for x in rng.integers(0, w["n"] + 1, int(rng.integers(0, 6)))]))
            for s0, s1 in zip(cuts[:-1], cuts[1:]):
                s.add_sequences(w["lengths"][s0:s1], w["codes"][w["offs"][s0]:w["offs"][s1]], first_seq=s0)
            stride = (w["n"] + 127) // 128 * 128
        got, _ = s.search(stride)
        again, _ = s.search(stride)                    # the resident copy (after a streamed first search)
    if chunks is not None:
        chunks.close()
    want, idx = oracle_matrix(w, go=go, ge=ge)
    assert len(idx) == w["n"]
    tag = f"seed {seed}: {w['n']} sequences, queries {w['m'].tolist()}, {w['matrix']} {go}/{ge}, {path}, options {opts}"
    assert np.array_equal(got[:, :w["n"]], want), tag
    assert np.array_equal(again[:, :w["n"]], want), tag + " (second search)"


def _small_db(rng):
    n = int(rng.choice([60, 129, 1200, 5000]))
    L = np.clip(rng.lognormal(np.log(float(rng.choice([40, 200]))), 0.6, n), 1, 2500).astype(np.int64)
    if rng.random() < 0.5:
        L[rng.integers(0, n)] = int(rng.integers(1500, 6000))
    L = np.sort(L).astype(np.uint16)
    total = int(L.astype(np.int64).sum())
    codes = rng.integers(0, 23, total).astype(np.int8)
    return L, codes, np.concatenate([[0], np.cumsum(L.astype(np.int64))])


def _batch(rng, L, codes, offs):
    nq = int(rng.choice([1, 2, 4, 10]))
    qlens = np.sort(np.clip(rng.lognormal(np.log(float(rng.choice([30, 300]))), 0.8, nq), 1, 1800).astype(np.int64))
    qs = []
    for ql in qlens:
        q = rng.integers(0, 23, int(ql)).astype(np.int8)
        i = int(rng.integers(0, len(L)))
        k = int(min(L[i], ql, 200))
        q[:k] = codes[offs[i]:offs[i] + k]
        qs.append(q)
    m = np.array([len(q) for q in qs], np.uint16)
    return np.concatenate(qs), m, np.concatenate([[0], np.cumsum(m.astype(np.int64))]).astype(np.uint32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWIMM_FUZZ_SESSIONS", "24"))))
def test_random_session_on_one_context(seed):
    """ONE context through a random sequence of calls -- new query batches, options flipped between searches, the database
    cleared and replaced (eager and lazy, chunks and slabs), whole vectors and top-r -- every result against the checker:
    what is cached between calls (work lists, launch plans, scratch, the uploader) must never outlive what it was built for."""
    from oracle import port
    rng = np.random.default_rng(5000 + seed)
    keep = []
    with hip_backend.HipSearcher(0) as s:
        L = codes = offs = None
        a = m = disp = None
        mat, go, ge = "blosum62", 10, 2
        chunks = None
        stride = 0
        for step in range(int(rng.integers(4, 9))):
            what = rng.choice(["db", "queries", "option", "search"]) if L is not None and a is not None else ("db" if L is None else "queries")
            if what == "db":
                L, codes, offs = _small_db(rng)
                s.clear_db()
                if chunks is not None:
                    keep.append(chunks)
                    chunks = None
                s.set_option("lazy_upload", int(rng.integers(0, 2)))
                if rng.random() < 0.5:
                    chunks = host.Chunks(L, codes, 128, int(rng.choice([3000, 40000, 1 << 20])))
                    for c in chunks.chunks:
                        s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
                    stride = chunks.vc * 128
                else:
                    cuts = sorted(set([0, len(L)] + [int(x) // 128 * 128 for x in rng.integers(0, len(L) + 1, int(rng.integers(0, 4)))]))
                    for s0, s1 in zip(cuts[:-1], cuts[1:]):
                        s.add_sequences(L[s0:s1], codes[offs[s0]:offs[s1]], first_seq=s0)
                    stride = (len(L) + 127) // 128 * 128
            elif what == "queries":
                if L is None:
                    continue
                a, m, disp = _batch(rng, L, codes, offs)
                mat, go, ge = str(rng.choice(MATRICES)), int(rng.integers(1, 16)), int(rng.integers(0, 4))
                s.set_queries(a, m, disp, submat.table(mat), go, ge)
            elif what == "option":
                key, vals = [("resident", [-1, 0, 1]), ("dynamic", [0, 1]), ("tail_mode", [0, 1, 2]), ("f16", [0, 1]), ("wg_limit", [0, 8]),
                             ("bnd_mib", [1, 16384]), ("score_mib", [1, 32768]), ("tail_frac", [10, 30]), ("tail_cap", [0, 25]),
                             ("sp_threshold", [0, 100, 65536]), ("stack", [0, 1]), ("cut", [0, 35]), ("max_waves", [0, 4])][int(rng.integers(0, 13))]
                s.set_option(key, int(rng.choice(vals)))
                continue
            if L is None or a is None:
                continue
            w = {"lengths": L, "codes": codes, "offs": offs, "n": len(L), "residues": int(offs[-1]), "a": a, "m": m, "disp": disp,
                 "query_residues": int(m.astype(np.int64).sum()), "matrix": mat}
            want, idx = oracle_matrix(w, go=go, ge=ge)
            tag = f"session {seed} step {step} ({what})"
            if rng.random() < 0.5:
                got, _ = s.search(stride)
                assert np.array_equal(got[:, :len(L)], want), tag
            else:
                r = int(rng.choice([1, 5, 20, 64, 100]))
                ts, ti, _ = s.search_topr(r, len(L))
                for k in range(len(m)):
                    os_, oi = port.topr(want[k], r)
                    assert np.array_equal(ts[k][:len(os_)], os_) and np.array_equal(ti[k][:len(oi)], oi), tag + f" top-{r} of query {k}"
    for c in keep + ([chunks] if chunks is not None else []):
        c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SWIMM_FUZZ_SHORT", "14"))))
def test_short_query_batches_share_workgroups(seed):
    """Batches of 8 to 240 SHORT queries (1 to 90 rows, a few longer ones among them): queries of up to 72 rows are stacked
    two to four to a workgroup (a zero boundary at every seam, one score row per member) -- in group-resident batch launches
    on a database that is small beside the chip, in rotation over three streams on a larger one -- and every member's
    whole score row must equal the checker's (a member of up to 72 rows cannot leave the binary16 tier's exact range: 72 x 17 < 1920)."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([300, 1500, 6000, 30000, 120000]))
    L = np.clip(rng.lognormal(np.log(float(rng.choice([40, 150, 300]))), 0.6, n), 1, 2500).astype(np.int64)
    if rng.random() < 0.3:
        L[rng.integers(0, n)] = int(rng.integers(2000, 8000))
    L = np.sort(L).astype(np.uint16)
    total = int(L.astype(np.int64).sum())
    codes = rng.integers(0, 23, total).astype(np.int8)
    offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
    nq = int(rng.choice([8, 9, 40, 100, 240]))
    qlens = rng.integers(1, 91, nq)
    for _ in range(int(rng.integers(0, 3))):
        qlens[rng.integers(0, nq)] = int(rng.integers(100, 900))
    qlens = np.sort(qlens)
    queries = []
    for ql in qlens:
        q = rng.integers(0, 23, int(ql)).astype(np.int8)
        if rng.random() < 0.5:
            i = int(rng.integers(0, n))
            k = int(min(L[i], ql))
            q[:k] = codes[offs[i]:offs[i] + k]
        queries.append(q)
    queries.sort(key=len)
    m = np.array([len(q) for q in queries], np.uint16)
    disp = np.concatenate([[0], np.cumsum(m.astype(np.int64))]).astype(np.uint32)
    w = {"lengths": L, "codes": codes, "offs": offs, "n": n, "residues": total, "a": np.concatenate(queries), "m": m, "disp": disp,
         "query_residues": int(m.astype(np.int64).sum()), "matrix": "blosum62" if seed % 2 == 0 else str(rng.choice(MATRICES))}
    go, ge = (10, 2) if seed % 2 == 0 else (int(rng.integers(0, 21)), int(rng.integers(0, 6)))
    opts = {}
    for key, vals in [("resident", [0, 1]), ("f16", [0]), ("tail_mode", [2]), ("score_mib", [1]), ("wg_limit", [8]), ("force_i32", [1])]:
        if rng.random() < 0.2:
            opts[key] = int(rng.choice(vals))
    sm = submat.table(w["matrix"])
    lazy = bool(rng.random() < 0.3)
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_option("lazy_upload", int(lazy))
        s.set_queries(w["a"], m, disp, sm, go, ge)
        s.add_sequences(L, codes, first_seq=0)
        stride = (n + 127) // 128 * 128
        got, _ = s.search(stride)
        s.set_option("stack", 0)
        plain, _ = s.search(stride)
    want, idx = oracle_matrix(w, go=go, ge=ge)
    tag = f"short-query seed {seed}: {n} sequences, {nq} queries {m.tolist()[:12]}..., {w['matrix']} {go}/{ge}, lazy {lazy}, options {opts}"
    assert np.array_equal(plain[:, :n], want), tag + " (stack=0)"
    assert np.array_equal(got[:, :n], want), tag
