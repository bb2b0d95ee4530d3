"""Edge cases of the search path on the GPU vs the CPU oracle: degenerate sizes, gap extremes, ragged
groups, the 16-bit length limits, and scores sitting exactly on the promotion thresholds
(2047/2048: binary16 tier, 32766/32767/32768: int16 tier)."""
import numpy as np
import pytest

from oracle import port
from swimm_amd import hip_backend, host, submat

pytestmark = pytest.mark.gpu

CODE = {c: i for i, c in enumerate("ABCDEFGHIKLMNPQRSTVWXYZ")}


def enc(s):
    return np.array([CODE[c] for c in s], dtype=np.int8)


def run_case(seqs, queries, matrix="blosum62", go=10, ge=2, opts=None, vl=128, max_chunk=1 << 20):
    seqs = sorted(seqs, key=len)                       # the database must be length-sorted (stable)
    lens = np.array([len(s) for s in seqs], dtype=np.uint16)
    codes = np.concatenate([s for s in seqs] + [np.zeros(0, np.int8)]).astype(np.int8)
    queries = sorted(queries, key=len)
    m = np.array([len(q) for q in queries], dtype=np.uint16)
    disp = np.concatenate([[0], np.cumsum(m)]).astype(np.uint32)
    a = np.concatenate(queries).astype(np.int8)
    sm = submat.table(matrix)
    ch = host.Chunks(lens, codes, vl, max_chunk)
    with hip_backend.HipSearcher(0) as s:
        for k, v in (opts or {}).items():
            s.set_option(k, v)
        s.set_queries(a, m, disp, sm, go, ge)
        for c in ch.chunks:
            s.add_chunk(c["b"], c["n"], c["disp"], vl, c["first_group"])
        got, _ = s.search(ch.vc * vl)
        stats = s.last_stats()
    ch.close()
    want = np.array([[port.pair_score(q, d, sm, go, ge) if len(d) else 0 for d in seqs] for q in queries], dtype=np.int32)
    assert np.array_equal(got[:, :len(seqs)], want), (got[:, :len(seqs)], want)
    return want, stats


def rnd(rng, n):
    return rng.integers(0, 23, n).astype(np.int8)


def test_degenerate_sizes():
    rng = np.random.default_rng(3)
    run_case([enc("W")], [enc("W")])                                   # 1 x 1
    run_case([enc("A"), enc("WW"), enc("")], [enc("W"), enc("AW")])    # empty record, length-1 and -2 queries
    run_case([rnd(rng, 5) for _ in range(129)], [rnd(rng, 33)])        # 128 + 1 sequences: second group almost empty
    run_case([rnd(rng, int(n)) for n in rng.integers(1, 40, 300)], [rnd(rng, 1), rnd(rng, 7), rnd(rng, 64)], vl=16, max_chunk=600)


@pytest.mark.parametrize("go,ge", [(0, 0), (0, 1), (1, 0), (127, 0), (100, 27), (5, 1)])
def test_gap_extremes(go, ge):
    rng = np.random.default_rng(go * 131 + ge)
    base = rnd(rng, 200)
    seqs = [rnd(rng, int(n)) for n in rng.integers(20, 400, 200)]
    seqs += [np.concatenate([base[:90], rnd(rng, 7), base[90:]]), np.delete(base, slice(50, 58))]   # homologs with one gap
    run_case(seqs, [base, base[:31]], go=go, ge=ge)
    run_case(seqs, [base], go=go, ge=ge, opts={"tail_mode": 1})


def test_length_limits():
    rng = np.random.default_rng(9)
    long_db = rnd(rng, 65535)                                           # longest sequence the .seq format can hold... minus rounding
    seqs = [rnd(rng, 50), rnd(rng, 51), long_db[:65530]]
    run_case(seqs, [long_db[1000:1030], rnd(rng, 5)])
    long_q = rnd(rng, 20001)                                            # 53 passes of the workgroup pipeline, 40 of the lane kernel
    seqs = [rnd(rng, int(n)) for n in rng.integers(1, 300, 140)] + [long_q[5000:5600].copy()]
    run_case(seqs, [long_q])
    run_case(seqs, [long_q], opts={"tail_mode": 2, "rows_per_wave": 16})


@pytest.mark.parametrize("opts", [{}, {"f16": 0}, {"tail_mode": 2}, {"tail_mode": 1}])
def test_scores_on_the_promotion_thresholds(opts):
    """self scores of W^k + filler under BLOSUM62 (W 11, C 9, H 8, A 4, K 5): exactly 2046..2049 and 32766..32768"""
    def mk(total):
        k, rest = divmod(total, 11)
        filler = {0: "", 1: None, 2: None, 3: None, 4: "A", 5: "K", 6: None, 7: "P", 8: "H", 9: "C", 10: "KK"}[rest]
        if filler is None:                     # trade one W (11) for letters that reach the remainder + 11
            k -= 1
            filler = {12: "AH", 13: "AC", 14: "PP", 17: "CH"}[rest + 11]
        return enc("W" * k + filler)
    targets = [2046, 2047, 2048, 2049, 32766, 32767, 32768]
    seqs = [mk(t) for t in targets]
    rng = np.random.default_rng(5)
    seqs += [rnd(rng, int(n)) for n in rng.integers(10, 200, 150)]
    want, stats = run_case(seqs, [mk(t) for t in targets], opts=opts)
    for t in targets:
        assert (want == t).any(), t            # every threshold value really occurs
    assert stats["promoted"] >= 2              # 32767 and 32768 needed the int32 tier


def _w_run(total):
    """a sequence whose self score under BLOSUM62 is exactly `total` (W 11, C 9, H 8, A 4, K 5, P 7)"""
    k, rest = divmod(total, 11)
    filler = {0: "", 1: None, 2: None, 3: None, 4: "A", 5: "K", 6: None, 7: "P", 8: "H", 9: "C", 10: "KK"}[rest]
    if filler is None:
        k -= 1
        filler = {12: "AH", 13: "AC", 14: "PP", 17: "CH"}[rest + 11]
    return enc("W" * k + filler)


@pytest.mark.parametrize("go,ge", [(10, 2), (5, 1), (11, 3), (4, 7), (12, 0), (3, 40)])
@pytest.mark.parametrize("opts", [{}, {"tail_mode": 2}, {"tail_mode": 2, "rows_per_wave": 20, "waves": 4, "resident": 1}])
def test_scores_around_the_binary16_tier_limit_for_any_extend_penalty(go, ge, opts):
    """the pipeline kernel's binary16 tier stores column j with (j mod P) * extend added and is exact below 2048 less the largest
    offset (sw_kernels.h, f16_exact_below): self scores on, just below and just above that limit and the old one, and homologs
    with gaps whose scores sit in between, for several extend penalties (each has its own period P and limit)"""
    period = 4 * (32 if ge <= 0 else max(1, 32 // ge))
    limit = 2048 - period * max(ge, 0)
    targets = sorted(set([limit - 13, limit - 3, limit - 2, limit - 1, limit, limit + 1, limit + 2, limit + 3, 2040, 2047, 2048, 2050]))
    seqs = [_w_run(t) for t in targets]
    rng = np.random.default_rng(ge + 31)
    for t in targets[:4]:                                # the same runs with an insertion and a deletion: gapped alignments near the limit
        w = _w_run(t + 40)
        seqs += [np.concatenate([w[:60], rnd(rng, 3), w[60:]]), np.delete(w, slice(100, 104))]
    seqs += [rnd(rng, int(n)) for n in rng.integers(10, 260, 150)]
    want, _ = run_case(seqs, [_w_run(t) for t in targets] + [_w_run(limit + 60)], go=go, ge=ge, opts=opts)
    for t in targets:
        assert (want == t).any(), t


@pytest.mark.parametrize("path", ["chunks", "slabs"])
@pytest.mark.parametrize("opts", [{}, {"tail_mode": 1}, {"tail_mode": 2, "f16": 0}, {"sp_threshold": 0}])
def test_every_residue_code_on_both_upload_paths(path, opts):
    """The device renumbers the residues when it tiles a chunk (retile_kernel) or a slab (tile_sequences_kernel, whose interior takes
    four residues per dword and whose edges go byte by byte) and the host builds every lookup table in that order: all 24 codes --
    the ambiguity codes and the dummy code 23 (J / O / U / *) included -- in the database and in the queries, against the
    checker, through the pipeline kernel, the lane-systolic kernel, the int16 tier and the score-profile kernel.  Bytes above 24
    in a database score like padding (0 against everything)."""
    rng = np.random.default_rng(77)
    seqs = [rng.integers(0, 24, int(n)).astype(np.int8) for n in rng.integers(1, 260, 300)]
    seqs += [np.arange(24, dtype=np.int8), np.tile(np.arange(24, dtype=np.int8)[::-1], 9), np.full(37, 23, np.int8)]
    queries = [np.arange(24, dtype=np.int8), np.tile(np.arange(24, dtype=np.int8), 5)[:101], rng.integers(0, 24, 333).astype(np.int8),
               np.concatenate([seqs[7], seqs[100]])[:150]]
    seqs = sorted(seqs, key=len)
    lens = np.array([len(x) for x in seqs], dtype=np.uint16)
    codes = np.concatenate(seqs).astype(np.int8)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    dirty = codes.copy()                                   # the same database with stray bytes: they must score 0, i.e. like a residue
    stray = rng.choice(len(codes), 40, replace=False)      # that matches nothing -- the checker sees code 24 (padding) there
    dirty[stray] = rng.choice(np.array([25, 31, 64, 127], np.int8), 40)
    clean_as_padding = codes.copy()
    clean_as_padding[stray] = 24
    queries = sorted(queries, key=len)
    m = np.array([len(q) for q in queries], dtype=np.uint16)
    disp = np.concatenate([[0], np.cumsum(m)]).astype(np.uint32)
    a = np.concatenate(queries).astype(np.int8)
    sm = submat.table("blosum50")
    for db, seen_by_checker in ((codes, codes), (dirty, clean_as_padding)):
        ch = None
        with hip_backend.HipSearcher(0) as s:
            for k, v in opts.items():
                s.set_option(k, v)
            s.set_queries(a, m, disp, sm, 11, 1)
            if path == "chunks":
                ch = host.Chunks(lens, db, 128, 30000)
                for c in ch.chunks:
                    s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
                stride = ch.vc * 128
            else:
                for s0, s1 in ((0, 128), (128, 256), (256, len(lens))):
                    s.add_sequences(lens[s0:s1], db[offs[s0]:offs[s1]], first_seq=s0)
                stride = (len(lens) + 127) // 128 * 128
            got, _ = s.search(stride)
        if ch is not None:
            ch.close()
        want = np.array([[port.pair_score(q, seen_by_checker[offs[i]:offs[i + 1]], sm, 11, 1) for i in range(len(lens))] for q in queries], dtype=np.int32)
        assert np.array_equal(got[:, :len(lens)], want), (path, opts, np.argwhere(got[:, :len(lens)] != want)[:5])


def test_chained_lane_passes_many_items():
    """every group through the lane-systolic kernel with 3 and 6 chained passes and thousands of items in
    flight: the inter-wave hand-over (boundary rows + progress counters through global memory) under load"""
    rng = np.random.default_rng(21)
    q1, q2 = rnd(rng, 1500), rnd(rng, 2900)
    seqs = [rnd(rng, int(n)) for n in rng.integers(1, 900, 2500)]
    seqs += [np.concatenate([rnd(rng, 40), q1[200:1300], rnd(rng, 15)]), np.concatenate([q2[:2000], rnd(rng, 300)])]
    for opts in ({"tail_mode": 1}, {"tail_mode": 1, "f16": 0}):
        want, stats = run_case(seqs, [q1, q2], matrix="blosum50", opts=opts, max_chunk=200000)
        assert want.max() > 2048


@pytest.mark.parametrize("T", [8, 12, 16, 20, 24, 28, 32, 36])
def test_every_strip_height_vs_oracle(T):
    """every rows-per-wave instantiation of the pipeline kernel, one pass and several, against the oracle;
    query lengths sit on and around the strip boundaries"""
    rng = np.random.default_rng(T)
    seqs = [rnd(rng, int(n)) for n in rng.integers(1, 300, 260)]
    base = rnd(rng, 4 * T + 3)
    seqs += [base.copy(), np.concatenate([base[:T], rnd(rng, 5), base[T:]])]
    queries = [base[:T - 1], base[:T], base[:T + 1], base[:2 * T], base[:3 * T + 2], base]
    for W in (1, 3, 4):
        run_case(seqs, queries, opts={"rows_per_wave": T, "waves": W, "tail_mode": 2})
    run_case(seqs, queries, opts={"rows_per_wave": T})
    run_case(seqs, queries, opts={"rows_per_wave": T, "waves": 4, "resident": 1})      # the group-resident instantiation of every height


def test_streamed_ranges_rotating_tail_launches_then_promotion():
    """A database that streams in as several ranges, small beside one extreme sequence (so the search is chain-bound and
    the tail launches of consecutive queries rotate over three streams), with scores >= 2048 on that sequence: the
    promotion ladder of a query must wait for the tail kernel of the FIRST range, whichever stream it ran on."""
    rng = np.random.default_rng(414)
    queries = [np.concatenate([np.full(230, CODE["W"], np.int8), rnd(rng, int(n))]) for n in (300, 340, 380, 420, 460, 500)]
    long_seq = np.concatenate([rnd(rng, 9000)] + [np.concatenate([q, rnd(rng, 2500)]) for q in queries] + [rnd(rng, 3000)])
    seqs = [rnd(rng, int(n)) for n in rng.integers(40, 260, 2600)] + [long_seq]
    opts = {"lazy_upload": 1, "resident": 0, "upload_piece_kib": 16}
    want, stats = run_case(seqs, queries, opts=opts, max_chunk=60000)
    assert (want[:, -1] >= 2048).all()            # every query leaves the binary16 range on the extreme sequence
    run_case(seqs, queries, opts=dict(opts, f16=0), max_chunk=60000)


def test_outlier_pairs_leave_their_group():
    """Groups whose last few sequences are far longer than the rest (Swiss-Prot's long end): the pipeline kernel stops at the
    longest pair that stays, the outlier pairs run whole through the lane-systolic kernel as well, and the two results merge
    in the score row -- one-pass and multi-pass queries, per-pass and group-resident launches, with hits that lie wholly
    beyond the cut, across it, and before it."""
    rng = np.random.default_rng(1234)
    q1, q2, q3 = rnd(rng, 90), rnd(rng, 700), rnd(rng, 1500)
    seqs = [rnd(rng, int(n)) for n in rng.integers(150, 400, 900)]
    # outliers: 3 to 40 times their neighbours' length, with copies of the queries planted at the start, across column ~400 and at the far end
    for k, n in enumerate([1200, 1300, 2500, 2600, 6000, 6100, 15000, 15001]):
        s = rnd(rng, n)
        src = (q1, q2, q3)[k % 3]
        at = (0, 380, n - len(src) - 7)[k % 3] if n > len(src) + 400 else 0
        s[at:at + len(src)] = src
        seqs.append(s)
    seqs += [rnd(rng, int(n)) for n in rng.integers(600, 700, 130)]          # a full group of medium sequences between bulk and outliers
    for opts in ({}, {"cut": 0}, {"cut": 5}, {"resident": 0}, {"resident": 1}, {"tail_mode": 2}, {"f16": 0}, {"lazy_upload": 1, "upload_piece_kib": 64}):
        want, stats = run_case(seqs, [q1, q2, q3], matrix="blosum50", opts=opts, max_chunk=300000)
    assert want.max() > 2048


def test_mass_promotion():
    """a family of 66 000 near-copies of the query: every alignment leaves the binary16 range (>= 2048), more than
    any fixed-size list would hold; plus 40 W-runs that also leave int16.  All of them come back exact."""
    rng = np.random.default_rng(77)
    base = rnd(rng, 450)
    fam = np.tile(base, (66000, 1))
    pos = rng.integers(0, 450, 66000)
    fam[np.arange(66000), pos] = rng.integers(0, 20, 66000)          # one substitution each
    seqs = [fam[i] for i in range(66000)]
    seqs += [np.full(int(n), CODE["W"], np.int8) for n in rng.integers(3000, 3100, 40)]
    wq = np.full(3080, CODE["W"], np.int8)
    seqs = sorted(seqs, key=len)
    lens = np.array([len(s) for s in seqs], dtype=np.uint16)
    codes = np.concatenate(seqs).astype(np.int8)
    queries = [base, wq]
    m = np.array([len(q) for q in queries], dtype=np.uint16)
    disp = np.concatenate([[0], np.cumsum(m)]).astype(np.uint32)
    a = np.concatenate(queries).astype(np.int8)
    sm = submat.table("blosum62")
    ch = host.Chunks(lens, codes, 128, 8 << 20)
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(a, m, disp, sm, 10, 2)
        for c in ch.chunks:
            s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
        got, _ = s.search(ch.vc * 128)
        st = s.last_stats()
    one = port.assemble_single_chunk(lens, codes, 128, 5)
    want = port.search_exact(a, m, disp, one["b"], one["n"], one["disp"], sm, 10, 2, 128)
    ch.close()
    n = len(seqs)
    assert np.array_equal(got[:, :n], want[:, :n])
    assert (want[0, :n] >= 2048).sum() > 65536 and st["promoted"] >= 40


def test_random_small_cases():
    """120 seeded random searches (database shape, queries, matrix, gaps, launch options) against the exact oracle"""
    rng = np.random.default_rng(20260104)
    mats = ["blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250"]
    for case in range(120):
        n_seq = int(rng.integers(1, 400))
        hi = int(rng.choice([12, 70, 300, 900]))
        lens = rng.integers(0 if rng.random() < 0.2 else 1, hi + 1, n_seq)
        seqs = [rnd(rng, int(n)) for n in lens]
        nq = int(rng.integers(1, 4))
        qlen_hi = int(rng.choice([8, 40, 200, 700, 1300]))
        queries = [rnd(rng, int(rng.integers(1, qlen_hi + 1))) for _ in range(nq)]
        if rng.random() < 0.3 and n_seq > 3:          # plant a close relative so that high scores occur
            src = queries[-1]
            seqs[int(rng.integers(0, n_seq))] = np.concatenate([rnd(rng, 5), src, rnd(rng, 3)]).astype(np.int8)
        go, ge = int(rng.integers(0, 20)), int(rng.integers(0, 6))
        opts = {}
        if rng.random() < 0.5:
            opts["rows_per_wave"] = int(rng.choice([8, 12, 16, 20, 24, 28, 32, 36]))
        if rng.random() < 0.5:
            opts["waves"] = int(rng.integers(1, 13))
        if rng.random() < 0.4:
            opts["tail_mode"] = int(rng.choice([1, 2]))
        if rng.random() < 0.2:
            opts["f16"] = 0
            opts.pop("rows_per_wave", None) if opts.get("rows_per_wave") not in (16, 24, 32) else None
        if rng.random() < 0.2:
            opts["dynamic"] = 0
        if rng.random() < 0.2:
            opts["bnd_mib"] = 1
        if rng.random() < 0.1:
            opts = {"force_i32": 1}
        try:
            run_case(seqs, queries, matrix=str(rng.choice(mats)), go=go, ge=ge, opts=opts, vl=int(rng.choice([16, 32, 64, 128])),
                     max_chunk=int(rng.choice([3000, 40000, 1 << 20])))
        except AssertionError as e:
            raise AssertionError(f"case {case}: n_seq={n_seq} queries={[len(q) for q in queries]} go={go} ge={ge} opts={opts}") from e


def test_many_queries():
    """300 queries in one call (events, cursors and profiles per query), with and without one-query batches"""
    rng = np.random.default_rng(5)
    seqs = [rnd(rng, int(n)) for n in rng.integers(1, 260, 700)]
    queries = [rnd(rng, int(n)) for n in rng.integers(1, 180, 300)]
    run_case(seqs, queries, opts={"tail_mode": 2})                  # short queries in rotation over three streams, no tail kernel
    run_case(seqs, queries[:60], opts={"score_mib": 0})
    run_case(seqs + [rnd(rng, 9000)], queries[:40])                 # an extreme sequence: the rotating queries keep their tail kernel
    run_case(seqs + [rnd(rng, 700)], queries[100:140] + [rnd(rng, 800)])   # short and multi-pass queries mixed
