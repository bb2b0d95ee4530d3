"""Pins the CPU oracle (oracle/sw_oracle.c + oracle/port.py) to the golden vectors that the
reference's own CPU path produced (tests/golden/make_golden.py), and -- where oracle/_ref is
present -- to the reference itself on fresh seeded inputs."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_npy
from oracle import port, ref
from swimm_amd import synth


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def small(tmp_path_factory, golden, oracle_built):
    tmp = tmp_path_factory.mktemp("oracle")
    prefix = str(tmp / "db")
    port.preprocess(os.path.join(GOLDEN, golden["db_fasta"]), prefix)
    pp = port.read_preprocessed(prefix)
    q = port.load_queries(os.path.join(GOLDEN, golden["query_fasta"]), 0)
    return {"prefix": prefix, "pp": pp, "q": q}


def test_preprocess_bytes(small, golden):
    g = golden["preprocess"]
    seq = open(small["prefix"] + ".seq", "rb").read()
    assert len(seq) == g["seq_bytes"] and sha(seq) == g["seq_sha256"]
    assert open(small["prefix"] + ".info").read() == g["info"]
    assert sha(open(small["prefix"] + ".desc", "rb").read()) == g["desc_clean_sha256"]
    assert np.array_equal(small["pp"]["lengths"], load_npy("db_small_lengths_sorted.npy"))


def test_query_layout(small, golden):
    g = golden["queries"]["mode0"]
    q = small["q"]
    assert sha(q["a"].tobytes()) == g["a_sha256"]
    assert q["m"].tolist() == g["m"] and q["lengths"].tolist() == g["lengths"]
    assert q["disp"].tolist() == g["disp"] and q["Q"] == g["Q"]
    assert q["titles"] == g["titles"]
    q1 = port.load_queries(os.path.join(GOLDEN, golden["query_fasta"]), 1)
    assert sha(q1["a"].tobytes()) == golden["queries"]["mode1"]["a_sha256"]
    assert q1["m"].tolist() == golden["queries"]["mode1"]["m"]


@pytest.mark.parametrize("vl,blk", [(32, 60), (16, 125)])
def test_single_chunk_layout(small, golden, vl, blk):
    g = golden["assemble"][f"single_vl{vl}_b{blk}"]
    a = port.assemble_single_chunk(small["pp"]["lengths"], small["pp"]["codes"], vl, blk)
    assert sha(a["b"].tobytes()) == g["b_sha256"]
    assert a["n"].tolist() == g["n"] and a["nbbs"].tolist() == g["nbbs"]
    assert [int(x) for x in a["disp"]] == g["disp"] and a["vD"] == g["vD"]


@pytest.mark.parametrize("vl,mx", [(16, 20000), (32, 50000)])
def test_multi_chunk_layout(small, golden, vl, mx):
    g = golden["assemble"][f"multi_vl{vl}_k{mx}"]
    a = port.assemble_multiple_chunks(small["pp"]["lengths"], small["pp"]["codes"], vl, mx)
    assert a["vc"] == g["vc"] and a["vD"] == g["vD"]
    assert [c["count"] for c in a["chunks"]] == g["counts"]
    assert [int(c["vD"]) for c in a["chunks"]] == g["chunk_vD"]
    assert [sha(c["b"].tobytes()) for c in a["chunks"]] == g["b_sha256"]
    assert [sha(c["disp"].tobytes()) for c in a["chunks"]] == g["disp_sha256"]


def _case_inputs(small, vl):
    pp, q = small["pp"], small["q"]
    a = port.assemble_single_chunk(pp["lengths"], pp["codes"], vl, 60)
    return q, a


def test_scores_exact_all_cases(small, golden):
    """exact int32 restatement == reference listing for every matrix / gap fixture."""
    if not ref.available():
        pytest.skip("matrix bytes come from oracle/_ref (golden holds only their hashes)")
    q, a = _case_inputs(small, 32)
    N = golden["search"]["n_sequences"]
    for name, c in golden["search"]["cases"].items():
        sm = ref.submat(c["matrix"])
        assert sha(sm.tobytes()) == golden["submat_sha256"][c["matrix"]]
        sc = port.search_exact(q["a"], q["m"], q["disp"], a["b"], a["n"], a["disp"], sm, c["open"], c["extend"], 32)
        want = load_npy(f"scores_{name}.npy")
        assert np.array_equal(sc[:, :N], want), name
        assert sc.max(axis=1).tolist() == c["max"]


def test_scores_tiered_and_lane_width(small, golden):
    """literal int8/int16/int32 tier restatement; result independent of the lane width."""
    if not ref.available():
        pytest.skip("needs oracle/_ref for the matrix bytes")
    N = golden["search"]["n_sequences"]
    c = golden["search"]["cases"]["blosum62_g10_e2"]
    sm = ref.submat("blosum62")
    want = load_npy("scores_blosum62_g10_e2.npy")
    q, a = _case_inputs(small, 32)
    sc, tiers = port.search_tiered(q["a"], q["m"], q["disp"], a["b"], a["n"], a["disp"], sm, 10, 2, 32, threads=4)
    assert np.array_equal(sc[:, :N], want)
    assert tiers[0] > 0 and tiers[1] > 0 and tiers[2] > 0  # every promotion tier fires
    q, a = _case_inputs(small, 128)
    sc = port.search_exact(q["a"], q["m"], q["disp"], a["b"], a["n"], a["disp"], sm, 10, 2, 128)
    assert np.array_equal(sc[:, :N], want)
    assert sc.max(axis=1).tolist() == c["max"]


def test_pair_scores_known_answers(oracle_built):
    """analytic pins: W^k self score = 11k under BLOSUM62 (needs only row/col 19 = W: 11)."""
    sm = np.zeros(768, dtype=np.int8)
    sm[19 * 32 + 19] = 11
    w = np.full(3200, 19, dtype=np.int8)
    assert port.pair_score(w, w, sm, 10, 2) == 35200
    assert port.pair_score(w[:7], w[:5], sm, 10, 2) == 55
    # one gap of length 2 between two 10-long matches: 110 + 110 - (10 + 2*2) = 206
    a = np.full(20, 19, dtype=np.int8)
    b = np.concatenate([w[:10], np.zeros(2, dtype=np.int8), w[:10]])
    assert port.pair_score(a, b, sm, 10, 2) == 206


def test_topr_order(golden, oracle_built):
    for name in golden["search"]["cases"]:
        sc = load_npy(f"scores_{name}.npy")
        order = load_npy(f"order_{name}.npy")
        for q in range(sc.shape[0]):
            s, i = port.topr(sc[q], sc.shape[1])
            assert np.array_equal(i, order[q]), (name, q)
            assert np.array_equal(s, sc[q][order[q]])
    # explicit tie rule: equal scores -> larger index first
    s, i = port.topr(np.array([5, 7, 5, 7, 1], dtype=np.int32), 4)
    assert s.tolist() == [7, 7, 5, 5] and i.tolist() == [3, 1, 2, 0]


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_against_reference_fresh_inputs(tmp_path, oracle_built):
    """reference vs restatement on inputs that are NOT in the fixtures (other seed, BLOSUM80)."""
    qs = synth.make_queries(77, [61, 250])
    db = synth.make_db(77, synth.lengths_lognormal(77, 300, 150, 0.6, 10, 900), planted=synth.planted_homologs(77, qs))
    dbfa, qfa = str(tmp_path / "d.fa"), str(tmp_path / "q.fa")
    synth.write_fasta(dbfa, synth.db_records(db), 73)
    synth.write_fasta(qfa, qs, 61)
    ref.preprocess_db(dbfa, str(tmp_path / "r"), 3)
    port.preprocess(dbfa, str(tmp_path / "p"))
    assert open(str(tmp_path / "r.seq"), "rb").read() == open(str(tmp_path / "p.seq"), "rb").read()
    assert open(str(tmp_path / "r.info"), "rb").read() == open(str(tmp_path / "p.info"), "rb").read()
    # the reference leaves one uninitialised byte after each title -- which can itself be a newline -- so the two files
    # are walked title by title instead of being split into lines
    rd = open(str(tmp_path / "r.desc"), "rb").read()
    pl = open(str(tmp_path / "p.desc"), "rb").read().split(b"\n")
    pos = 0
    for y in pl:
        if not y:
            continue
        assert rd[pos:pos + len(y)] == y, y
        pos += len(y)
        if rd[pos:pos + 2] == b"\n\n":
            pos += 2                                   # the stray byte happens to be a newline
        elif rd[pos:pos + 1] == b"\n":
            pos += 1                                   # no stray byte
        else:
            assert rd[pos + 1:pos + 2] == b"\n"
            pos += 2                                   # stray byte, then the newline
    assert pos == len(rd)
    rq, pq = ref.load_queries(qfa, 0, 2), port.load_queries(qfa, 0)
    for k in ("a", "m", "lengths", "disp"):
        assert np.array_equal(rq[k], pq[k]), k
    rs = ref.assemble_single_chunk(str(tmp_path / "r"), 32, 60, 1)
    pp = port.read_preprocessed(str(tmp_path / "p"))
    ps = port.assemble_single_chunk(pp["lengths"], pp["codes"], 32, 60)
    for k in ("b", "n", "nbbs", "disp"):
        assert np.array_equal(rs[k], ps[k]), k
    sm = ref.submat("blosum80")
    want, _ = ref.cpu_search(rq["a"], rq["m"], rq["disp"], rs["b"], rs["n"], rs["nbbs"], rs["disp"], sm, 11, 1, 32, threads=4)
    got = port.search_exact(pq["a"], pq["m"], pq["disp"], ps["b"], ps["n"], ps["disp"], sm, 11, 1, 32)
    assert np.array_equal(want, got)
    got_t, _ = port.search_tiered(pq["a"], pq["m"], pq["disp"], ps["b"], ps["n"], ps["disp"], sm, 11, 1, 32, threads=4)
    assert np.array_equal(want, got_t)
    n = rs["sequences_count"]
    for qi in range(2):
        s1, i1 = ref.sort_scores(want[qi, :n], 2)
        s2, i2 = port.topr(want[qi, :n], n)
        assert np.array_equal(s1, s2) and np.array_equal(i1, i2)
