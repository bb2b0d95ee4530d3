"""Product host library (swimm_amd/csrc/host, via swimm_amd.host) vs the golden vectors the
reference's own preprocess_db / load_query_sequences / assemble_*_db / sort_scores produced,
and vs the oracle restatement on ragged / edge inputs."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_npy
from oracle import port
from swimm_amd import host, submat, synth


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def small(tmp_path_factory, golden):
    tmp = tmp_path_factory.mktemp("host")
    prefix = str(tmp / "db")
    n, d = host.preprocess_db(os.path.join(GOLDEN, golden["db_fasta"]), prefix)
    return {"prefix": prefix, "n": n, "d": d, "db": host.db_load(prefix)}


def test_substitution_tables(golden):
    for name in submat.NAMES:
        assert sha(host.submat(name).tobytes()) == golden["submat_sha256"][name], name
        assert np.array_equal(host.submat(name), submat.table(name))
    with pytest.raises(host.SwimmHostError):
        host.submat("blosum99")


def test_alphabet():
    letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabjz*-", dtype=np.uint8)
    got = host.recode(letters)
    assert np.array_equal(got[:26], port.recode(letters[:26]))
    assert got[:26].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 23, 9, 10, 11, 12, 23, 13, 14, 15, 16, 17, 23, 18, 19, 20, 21, 22]
    assert got[26:].tolist() == [0, 1, 23, 22, 23, 23]   # lower case folds, junk -> dummy


def test_preprocess_bytes(small, golden):
    g = golden["preprocess"]
    seq = open(small["prefix"] + ".seq", "rb").read()
    assert len(seq) == g["seq_bytes"] and sha(seq) == g["seq_sha256"]
    assert open(small["prefix"] + ".info").read() == g["info"]
    assert sha(open(small["prefix"] + ".desc", "rb").read()) == g["desc_clean_sha256"]
    assert np.array_equal(small["db"]["lengths"], load_npy("db_small_lengths_sorted.npy"))
    assert (small["n"], small["d"]) == tuple(int(x) for x in g["info"].split()[:2])


@pytest.mark.parametrize("pad,key", [(True, "mode0"), (False, "mode1")])
def test_query_batch(golden, pad, key):
    g = golden["queries"][key]
    q = host.queries_load(os.path.join(GOLDEN, golden["query_fasta"]), pad)
    assert sha(q["a"].tobytes()) == g["a_sha256"] and q["m"].tolist() == g["m"] and q["Q"] == g["Q"]
    if pad:
        assert q["lengths"].tolist() == g["lengths"] and q["disp"].tolist() == g["disp"] and q["titles"] == g["titles"]


@pytest.mark.parametrize("vl,blk", [(32, 60), (16, 125)])
def test_single_chunk_layout(small, golden, vl, blk):
    g = golden["assemble"][f"single_vl{vl}_b{blk}"]
    a = host.assemble_single_chunk(small["db"]["lengths"], small["db"]["codes"], vl, blk)
    assert sha(a["b"].tobytes()) == g["b_sha256"] and a["vD"] == g["vD"]
    assert a["n"].tolist() == g["n"] and a["nbbs"].tolist() == g["nbbs"] and [int(x) for x in a["disp"]] == g["disp"]


@pytest.mark.parametrize("vl,mx", [(16, 20000), (32, 50000)])
def test_chunked_layout(small, golden, vl, mx):
    g = golden["assemble"][f"multi_vl{vl}_k{mx}"]
    ch = host.Chunks(small["db"]["lengths"], small["db"]["codes"], vl, mx)
    assert ch.vc == g["vc"] and ch.vD == g["vD"]
    assert [c["count"] for c in ch.chunks] == g["counts"] and [c["vD"] for c in ch.chunks] == g["chunk_vD"]
    assert [sha(c["b"].tobytes()) for c in ch.chunks] == g["b_sha256"]
    assert [sha(c["disp"].tobytes()) for c in ch.chunks] == g["disp_sha256"]
    assert [c["first_group"] for c in ch.chunks] == np.concatenate([[0], np.cumsum(g["counts"])[:-1]]).tolist()
    ch.close()


def test_layout_lane128_vs_oracle(small):
    """the lane width the GPU path uses (128) is not in the reference's fixtures: check vs the oracle"""
    db = small["db"]
    want = port.assemble_multiple_chunks(db["lengths"].astype(np.int64), db["codes"], 128, 30000)
    ch = host.Chunks(db["lengths"], db["codes"], 128, 30000)
    assert len(ch.chunks) == len(want["chunks"])
    for a, b in zip(ch.chunks, want["chunks"]):
        assert np.array_equal(a["b"], b["b"]) and np.array_equal(a["n"], b["n"]) and np.array_equal(a["disp"], b["disp"])
    ch.close()


def test_topr_order(golden):
    for name in golden["search"]["cases"]:
        sc, order = load_npy(f"scores_{name}.npy"), load_npy(f"order_{name}.npy")
        for q in range(sc.shape[0]):
            for r in (1, 10, sc.shape[1], sc.shape[1] + 5):
                s, i = host.topr(sc[q], r)
                k = min(r, sc.shape[1])
                assert np.array_equal(i[:k], order[q][:k]) and np.array_equal(s[:k], sc[q][order[q][:k]])
                assert (i[k:] == -1).all() and (s[k:] == -1).all()
    s, i = host.topr(np.array([5, 7, 5, 7, 1], dtype=np.int32), 4)
    assert s.tolist() == [7, 7, 5, 5] and i.tolist() == [3, 1, 2, 0]


def test_topr_merge_equals_global(golden):
    sc = load_npy("scores_blosum62_g10_e2.npy")[2]
    n, r = len(sc), 20
    bounds = [0, 100, 101, 260, n]
    ls, li = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s, i = host.topr(sc[a:b], r)
        ls.append(s); li.append(np.where(i >= 0, i + a, -1))
    ms, mi = host.topr_merge(np.stack(ls), np.stack(li), r)
    ws, wi = host.topr(sc, r)
    assert np.array_equal(ms, ws) and np.array_equal(mi, wi)


def test_titles_lookup(small, golden):
    idx = np.array([5, 0, small["n"] - 1, 5], dtype=np.int64)
    t = host.db_titles(small["prefix"], small["n"], idx)
    lines = open(small["prefix"] + ".desc").read().split("\n")
    assert t == [lines[i][1:] for i in idx]


def test_titles_through_the_offset_sidecar(small, tmp_path):
    """<prefix>.didx (written by preprocess beside the reference's three files: "SWIMDIDX", N, size of .desc, N line offsets) turns a
    report's title lookups into positioned reads (load_database_headers reads all N, sequences.c:757-761); without it, or with a
    stale / damaged one, the file is walked -- the titles are the same either way."""
    import shutil
    import struct
    raw = open(small["prefix"] + ".didx", "rb").read()
    magic, n, size = struct.unpack("<QQQ", raw[:24])
    assert magic == int.from_bytes(b"SWIMDIDX", "little") and n == small["n"] and size == os.path.getsize(small["prefix"] + ".desc")
    off = np.frombuffer(raw[24:], dtype=np.uint64)
    desc = open(small["prefix"] + ".desc", "rb").read()
    assert len(off) == n and off[0] == 0 and all(desc[int(o) - 1:int(o)] == b"\n" for o in off[1:]) and desc[int(off[-1]):].count(b"\n") == 1
    idx = np.array([small["n"] - 1, 0, 17, 17, small["n"] // 2], dtype=np.int64)
    want = [desc.split(b"\n")[i][1:].decode("latin1") for i in idx]
    assert host.db_titles(small["prefix"], small["n"], idx) == want
    # the same database without the sidecar, with one that names another .desc size, and with a truncated one: the walk answers
    for k, damage in enumerate((None, "size", "short")):
        p = str(tmp_path / f"db{k}")
        for ext in (".desc", ".info", ".seq"):
            shutil.copy(small["prefix"] + ext, p + ext)
        if damage == "size":
            open(p + ".didx", "wb").write(raw[:16] + struct.pack("<Q", size + 1) + raw[24:])
        elif damage == "short":
            open(p + ".didx", "wb").write(raw[:-8])
        assert host.db_titles(p, small["n"], idx) == want, damage
    with pytest.raises(host.SwimmHostError):
        host.db_titles(small["prefix"], small["n"], np.array([small["n"]], dtype=np.int64))


def test_ragged_and_edge_inputs(tmp_path):
    """CRLF, blank lines, lower case, one-residue and empty records, no trailing newline"""
    fa = tmp_path / "odd.fa"
    fa.write_bytes(b">one\r\nACDE\r\nfgh\r\n\r\n>two empty\n>three\nW\n>four\nMKV\nLLA")
    n, d = host.preprocess_db(str(fa), str(tmp_path / "odd"))
    assert (n, d) == (4, 14)
    db = host.db_load(str(tmp_path / "odd"))
    assert db["lengths"].tolist() == [0, 1, 6, 7]
    assert open(str(tmp_path / "odd.desc")).read() == ">two empty\n>three\n>four\n>one\n"
    assert db["codes"].tolist() == [19] + port.recode(np.frombuffer(b"MKVLLA", np.uint8)).tolist() + port.recode(np.frombuffer(b"ACDEFGH", np.uint8)).tolist()
    with pytest.raises(host.SwimmHostError) as e:
        host.queries_load(str(fa), True)   # empty query is an error, not UB
    assert e.value.status == 4
    with pytest.raises(host.SwimmHostError) as e:
        host.preprocess_db(str(tmp_path / "missing.fa"), str(tmp_path / "x"))
    assert e.value.status == 2   # the reference's exit(2)
    long = tmp_path / "long.fa"
    long.write_bytes(b">too long\n" + b"A" * 70000 + b"\n")
    with pytest.raises(host.SwimmHostError):
        host.preprocess_db(str(long), str(tmp_path / "long"))


def test_large_random_layout_vs_oracle():
    L = np.sort(synth.lengths_lognormal(9, 3000, 120, 0.8, 1, 3000)).astype(np.uint16)
    codes = port.recode(synth.residues(9, 3, 0, int(L.sum())))
    want = port.assemble_single_chunk(L.astype(np.int64), codes, 64, 60)
    got = host.assemble_single_chunk(L, codes, 64, 60)
    for k in ("b", "n", "nbbs", "disp"):
        assert np.array_equal(got[k], want[k]), k
