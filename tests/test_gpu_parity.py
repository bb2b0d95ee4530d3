"""GPU parity tests proper: HIP path (through the C-ABI) vs golden vectors and vs the CPU oracle.
Bit-exact: these are integer scores."""
import numpy as np
import pytest

from conftest import load_npy
from helpers import golden_inputs, load_chunks, make_chunks, matrix
from oracle import port
from swimm_amd import hip_backend, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def searcher():
    s = hip_backend.HipSearcher(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def gin(tmp_path_factory, golden):
    return golden_inputs(tmp_path_factory.mktemp("gpu"), golden, vl=128)


def _run(searcher, q, chunked, vl, sm, go, ge):
    searcher.clear_db()
    searcher.set_queries(q["a"], q["m"], q["disp"], sm, go, ge)
    vc = load_chunks(searcher, chunked, vl)
    scores, wt = searcher.search(vc * vl)
    return scores


def test_golden_all_cases(searcher, gin, golden):
    """every matrix / gap fixture; includes multi-pass (3200-row query) and the int32 promotion tier"""
    q, pp, chunked = gin
    N = golden["search"]["n_sequences"]
    for name, c in golden["search"]["cases"].items():
        sc = _run(searcher, q, chunked, 128, matrix(c["matrix"]), c["open"], c["extend"])
        want = load_npy(f"scores_{name}.npy")
        assert np.array_equal(sc[:, :N], want), name
        st = searcher.last_stats()
        assert st["promoted"] > 0, "W x 3200 self hit must go through the int32 tier"


@pytest.mark.parametrize("opts", [{"force_i32": 1}, {"rows_per_wave": 16}, {"max_waves": 1}, {"max_waves": 5, "wg_limit": 256},
                                  {"tail_mode": 1}, {"tail_mode": 2}, {"tail_mode": 1, "rows_per_wave": 16}, {"rows_per_wave": 24},
                                  {"f16": 0}, {"f16": 0, "tail_mode": 2}, {"f16": 1, "rows_per_wave": 24, "tail_mode": 2},
                                  {"dynamic": 0}, {"dynamic": 0, "tail_mode": 2, "f16": 0}, {"dynamic": 0, "tail_mode": 2, "max_waves": 3},
                                  {"rows_per_wave": 8}, {"rows_per_wave": 12, "tail_mode": 2}, {"rows_per_wave": 20}, {"rows_per_wave": 28, "waves": 4},
                                  {"rows_per_wave": 36}, {"rows_per_wave": 36, "waves": 3, "tail_mode": 2}, {"rows_per_wave": 20, "waves": 7},
                                  {"score_mib": 0}, {"score_mib": 0, "tail_mode": 2}, {"resident": 0, "bnd_mib": 1, "tail_mode": 2},
                                  {"resident": 0, "bnd_mib": 1, "tail_mode": 2, "rows_per_wave": 16, "waves": 4}, {"resident": 0, "bnd_mib": 1, "tail_mode": 2, "force_i32": 1},
                                  {"tail_mode": 2, "rows_per_wave": 12, "waves": 3},
                                  # one launch per query and pass (the default for a single query)
                                  {"resident": 0}, {"resident": 0, "tail_mode": 2}, {"resident": 0, "tail_mode": 1},
                                  {"resident": 0, "rows_per_wave": 28, "waves": 8, "tail_mode": 2},
                                  # group-resident passes (one launch per multi-pass query), also with a deep pipeline: most groups
                                  # are then shorter than the pipeline and idle between their passes
                                  {"resident": 1}, {"resident": 1, "tail_mode": 2}, {"resident": 1, "tail_mode": 1},
                                  {"resident": 1, "tail_mode": 2, "rows_per_wave": 8, "waves": 16}, {"resident": 1, "tail_mode": 2, "rows_per_wave": 28, "waves": 8}, {"resident": 1, "tail_mode": 2, "rows_per_wave": 28, "waves": 16},
                                  {"resident": 1, "tail_mode": 2, "rows_per_wave": 36, "waves": 12}, {"resident": 1, "f16": 0, "tail_mode": 2, "rows_per_wave": 16, "waves": 5},
                                  {"resident": 1, "force_i32": 1, "rows_per_wave": 16, "waves": 3}])
def test_golden_kernel_variants(gin, golden, opts):
    q, pp, chunked = gin
    N = golden["search"]["n_sequences"]
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items():
            s.set_option(k, v)
        sc = _run(s, q, chunked, 128, matrix("blosum62"), 10, 2)
    assert np.array_equal(sc[:, :N], load_npy("scores_blosum62_g10_e2.npy"))


@pytest.mark.parametrize("vl,max_chunk", [(16, 20000), (32, 50000), (64, None), (128, 30000)])
def test_reference_chunk_layouts(searcher, tmp_path, golden, vl, max_chunk):
    """the boundary accepts the reference's own chunk layout for any lane width dividing 128"""
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=vl, max_chunk=max_chunk)
    N = golden["search"]["n_sequences"]
    sc = _run(searcher, q, chunked, vl, matrix("pam250"), 10, 2)
    assert np.array_equal(sc[:, :N], load_npy("scores_pam250_g10_e2.npy"))


def test_seeded_db_vs_oracle(searcher):
    """5 000 log-normal sequences x 3 queries vs the exact CPU restatement"""
    qs = synth.make_queries(5, [97, 375, 1000])
    db = synth.make_db(5, synth.lengths_lognormal(5, 5000, 250, 0.7, 5, 4000), planted=synth.planted_homologs(5, qs), with_titles=False)
    order = port.stable_sort_by_length(db.lengths)
    offs = np.concatenate([[0], np.cumsum(db.lengths)])
    lens = db.lengths[order]
    codes = np.concatenate([port.recode(db.letters[offs[i]:offs[i + 1]]) for i in order])
    qa = [port.recode(s) for _, s in qs]
    m = np.array([len(x) for x in qa], dtype=np.uint16)
    disp = np.concatenate([[0], np.cumsum(m)]).astype(np.uint32)
    a = np.concatenate(qa)
    chunked = make_chunks(lens, codes, 128, 200000)
    sm = matrix("blosum50")
    searcher.clear_db()
    searcher.set_queries(a, m, disp, sm, 10, 2)
    vc = load_chunks(searcher, chunked, 128)
    got, _ = searcher.search(vc * 128)
    one = port.assemble_single_chunk(lens, codes, 128, 5)
    want = port.search_exact(a, m, disp, one["b"], one["n"], one["disp"], sm, 10, 2, 128)
    assert np.array_equal(got, want)
    # top-r through the ABI == the reference's sorted listing order
    n = len(lens)
    for r in (1, 25, 64, 100):   # <= 64: device wave-shuffle top-r; > 64: host selection
        ts, ti, _ = searcher.search_topr(r, n)
        for qi in range(3):
            s, i = port.topr(want[qi, :n], r)
            assert np.array_equal(ts[qi], s) and np.array_equal(ti[qi], i), (r, qi)
    # one query per batch (score-row budget) and a boundary buffer cut into runs: same answers
    searcher.set_option("score_mib", 0)
    searcher.set_option("bnd_mib", 1)
    got2, _ = searcher.search(vc * 128)
    assert np.array_equal(got2, want)
    for r in (7, 80):
        ts, ti, _ = searcher.search_topr(r, n)
        for qi in range(3):
            s, i = port.topr(want[qi, :n], r)
            assert np.array_equal(ts[qi], s) and np.array_equal(ti[qi], i), (r, qi)
    assert searcher.last_plan(2)["rows_per_wave"] > 0
    searcher.set_option("score_mib", 32768)
    searcher.set_option("bnd_mib", 16384)
    # n_valid below the resident count: padding / excluded lanes never show up
    ts, ti, _ = searcher.search_topr(10, n - 77)
    for qi in range(3):
        s, i = port.topr(want[qi, :n - 77], 10)
        assert np.array_equal(ts[qi], s) and np.array_equal(ti[qi], i)


def test_search_chunks_dropin(tmp_path, golden):
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=16, max_chunk=20000)
    N = golden["search"]["n_sequences"]
    sc, wt = hip_backend.search_chunks(q["a"], q["m"], q["disp"], chunked["vc"], chunked["chunks"], matrix("blosum62"), 10, 2, 1, 16)
    assert np.array_equal(sc[:, :N], load_npy("scores_blosum62_g10_e2.npy")) and wt > 0


@pytest.mark.parametrize("cut", [None, 128, 256, 384])
def test_add_sequences_direct_path(searcher, gin, golden, cut):
    """the .seq content (lengths + concatenated codes) tiled on the device, in one slab or two: same scores as the
    reference chunk layout"""
    q, pp, chunked = gin
    N = golden["search"]["n_sequences"]
    lens, codes = pp["lengths"], pp["codes"]
    searcher.clear_db()
    searcher.set_queries(q["a"], q["m"], q["disp"], matrix("blosum62"), 10, 2)
    if cut is None:
        searcher.add_sequences(lens, codes, 0)
    else:
        r = int(lens[:cut].astype(np.int64).sum())
        searcher.add_sequences(lens[:cut], codes[:r], 0)
        searcher.add_sequences(lens[cut:], codes[r:], cut)
    stride = (N + 127) // 128 * 128 + 128
    sc, _ = searcher.search(stride)
    assert np.array_equal(sc[:, :N], load_npy("scores_blosum62_g10_e2.npy"))
    ts, ti, _ = searcher.search_topr(30, N)
    for qi in range(sc.shape[0]):
        s, i = port.topr(sc[qi, :N], 30)
        assert np.array_equal(ts[qi], s) and np.array_equal(ti[qi], i)


@pytest.mark.parametrize("gpus", [2, 3])
def test_search_chunks_shards_over_devices(tmp_path, golden, monkeypatch, gpus):
    """the multi-GPU host path of the drop-in call (static shard, one thread and one context per device, scatter into
    one score array) on the one-GPU test box: SWIMM_HIP_VIRTUAL_GPUS maps device d to physical device d % 1"""
    monkeypatch.setenv("SWIMM_HIP_VIRTUAL_GPUS", str(gpus))
    assert hip_backend.device_count() == gpus
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=32, max_chunk=12000)
    assert len(chunked["chunks"]) >= gpus
    N = golden["search"]["n_sequences"]
    sc, wt = hip_backend.search_chunks(q["a"], q["m"], q["disp"], chunked["vc"], chunked["chunks"], matrix("blosum62"), 10, 2, gpus, 32)
    assert np.array_equal(sc[:, :N], load_npy("scores_blosum62_g10_e2.npy"))
    with pytest.raises(hip_backend.SwimmHipError):
        hip_backend.search_chunks(q["a"], q["m"], q["disp"], chunked["vc"], chunked["chunks"], matrix("blosum62"), 10, 2, gpus + 1, 32)


def test_errors_are_loud():
    with hip_backend.HipSearcher(0) as s:
        with pytest.raises(hip_backend.SwimmHipError):
            s.search(128)  # nothing set
        with pytest.raises(hip_backend.SwimmHipError):
            s.set_queries(np.zeros(4, np.int8), np.array([4], np.uint16), np.array([0, 4], np.uint32), np.zeros(768, np.int8), 100, 100)
        with pytest.raises(hip_backend.SwimmHipError):
            s.add_chunk(np.zeros(48, np.int8), np.array([1], np.uint16), np.array([0], np.uint32), 48, 0)


@pytest.mark.parametrize("opts", [{}, {"tail_mode": 2}, {"tail_mode": 1}, {"dynamic": 0}, {"f16": 0, "bnd_mib": 1}, {"rows_per_wave": 16, "waves": 4},
                                  {"resident": 0}, {"resident": 1}, {"resident": 0, "tail_mode": 2, "wg_limit": 8}])
def test_streaming_upload_same_scores(tmp_path, golden, opts):
    """X2 overlapped with compute (MICsearch.c:85-91): with "lazy_upload" the chunks are copied and tiled while earlier
    chunks are being aligned, each chunk with work lists of its own.  Same golden scores with uploads in flight, in the
    reference chunk layout (6 chunks) and as .seq slabs; the second search runs on the resident copy."""
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=128, max_chunk=16000)
    assert len(chunked["chunks"]) >= 4
    N = golden["search"]["n_sequences"]
    want = load_npy("scores_blosum62_g10_e2.npy")
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_option("lazy_upload", 1)
        s.set_queries(q["a"], q["m"], q["disp"], matrix("blosum62"), 10, 2)
        vc = load_chunks(s, chunked, 128)
        first, _ = s.search(vc * 128)
        st = s.last_stats()
        again, _ = s.search(vc * 128)
        assert np.array_equal(first[:, :N], want) and np.array_equal(again[:, :N], want)
        assert st["promoted"] > 0
        # .seq slabs of 128 sequences, streamed the same way
        s.clear_db()
        lens, codes = pp["lengths"].astype(np.uint16), pp["codes"]
        offs = np.concatenate([[0], np.cumsum(pp["lengths"])])
        for s0 in range(0, N, 128):
            s1 = min(N, s0 + 128)
            s.add_sequences(lens[s0:s1], codes[offs[s0]:offs[s1]], first_seq=s0)
        slab, _ = s.search((N + 127) // 128 * 128)
        ts, ti, _ = s.search_topr(10, N)
        assert np.array_equal(slab[:, :N], want)
        for k in range(want.shape[0]):
            os_, oi = port.topr(want[k], 10)
            assert np.array_equal(ts[k], os_) and np.array_equal(ti[k], oi)


def test_streaming_upload_c2_shape():
    """a tenth of the c2 shard in 12 chunks: streamed search == search of the resident copy == eager upload, and the
    whole-call drop-in (always streaming) agrees"""
    import bench
    from swimm_amd import host, submat
    shard = bench.build_shard(2, 0.1)
    chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 5 << 20)
    assert len(chunks.chunks) >= 10
    qa = shard["query"]
    m, disp, sm = np.array([len(qa)], np.uint16), np.array([0, len(qa)], np.uint32), submat.table("blosum62")
    res = {}
    for lazy in (0, 1):
        with hip_backend.HipSearcher(0) as s:
            s.set_option("lazy_upload", lazy)
            s.set_queries(qa, m, disp, sm, 10, 2)
            for ch in chunks.chunks:
                s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
            res[lazy], _ = s.search(chunks.vc * 128)
            if lazy:
                res["again"], _ = s.search(chunks.vc * 128)
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res["again"])
    allsc, _ = hip_backend.search_chunks(qa, m, disp, chunks.vc, chunks.chunks, sm, 10, 2, 1, 128)
    assert np.array_equal(allsc, res[0])
    offs = np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))])
    for i in list(np.random.default_rng(3).integers(0, shard["n"], 60)) + [int(np.argmax(res[0][0, :shard["n"]]))]:
        assert res[0][0, i] == port.pair_score(qa, shard["codes"][offs[i]:offs[i + 1]], sm, 10, 2), i
    chunks.close()


@pytest.mark.parametrize("opts", [{}, {"wg_limit": 3}, {"upload_piece_kib": 16}, {"upload_piece_kib": 16, "wg_limit": 2, "rows_per_wave": 24},
                                  {"f16": 0}, {"max_waves": 4}, {"tail_mode": 1}])
def test_one_query_walks_one_list_while_the_database_lands(tmp_path, golden, opts):
    """A streaming search for ONE query that fits one pass is ONE pipeline launch over one item list in upload order; its
    workgroups wait for the parts as they land (PipeParams::avail, publish_items_kernel).  Every golden query on its own, many
    small parts, two or three workgroups that take everything (each waits for every part), the promotion ladder behind it:
    the golden scores, and one launch where the path applies."""
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=128, max_chunk=9000)
    N = golden["search"]["n_sequences"]
    want = load_npy("scores_blosum62_g10_e2.npy")
    lens, codes = pp["lengths"].astype(np.uint16), pp["codes"]
    offs = np.concatenate([[0], np.cumsum(pp["lengths"])])
    for k in range(len(q["m"])):
        a = q["a"][q["disp"][k]:q["disp"][k] + q["m"][k]]
        m, disp = np.array([q["m"][k]], np.uint16), np.array([0, q["m"][k]], np.uint32)
        for path in ("chunks", "slabs"):
            with hip_backend.HipSearcher(0) as s:
                for key, v in opts.items():
                    s.set_option(key, v)
                s.set_option("lazy_upload", 1)
                s.set_queries(a, m, disp, matrix("blosum62"), 10, 2)
                if path == "chunks":
                    stride = load_chunks(s, chunked, 128) * 128
                else:
                    for s0 in range(0, N, 256):
                        s1 = min(N, s0 + 256)
                        s.add_sequences(lens[s0:s1], codes[offs[s0]:offs[s1]], first_seq=s0)
                    stride = (N + 127) // 128 * 128
                first, _ = s.search(stride)
                st, plan = s.last_stats(), s.last_plan(0)
                again, _ = s.search(stride)
            assert np.array_equal(first[0, :N], want[k]) and np.array_equal(again[0, :N], want[k]), (k, path, opts)
            if plan["passes"] == 1 and opts.get("f16", 1) and int(q["m"][k]) <= 400:
                assert st["launches"] <= 1 + 2 * 2, (k, path, opts, st)      # the one launch (+ the ladder's re-runs), not one per range


def test_two_contexts_interleaved_on_one_thread(tmp_path, golden, monkeypatch):
    """Two live contexts on two (virtual) devices driven by ONE host thread in interleaved order -- create A, create B,
    chunks into A, chunks into B, search B, search A, top-r A, more chunks into B, search B: every entry point has to make
    its own context's device current again (the library records the device last entered per thread and refuses a HIP call
    issued for another context: a missing switch is an error here even though both devices are one physical GPU)."""
    monkeypatch.setenv("SWIMM_HIP_VIRTUAL_GPUS", "2")
    q, pp, chunked = golden_inputs(tmp_path, golden, vl=128, max_chunk=30000)
    N = golden["search"]["n_sequences"]
    want = load_npy("scores_blosum62_g10_e2.npy")
    sm = matrix("blosum62")
    chunks = chunked["chunks"]
    firsts = np.concatenate([[0], np.cumsum([c["count"] for c in chunks])])
    half = len(chunks) // 2
    assert half >= 1
    A, B = hip_backend.HipSearcher(0), hip_backend.HipSearcher(1)
    try:
        A.set_queries(q["a"], q["m"], q["disp"], sm, 10, 2)
        B.set_option("lazy_upload", 1)                   # B streams its chunks in with its own uploader thread
        B.set_queries(q["a"], q["m"], q["disp"], sm, 10, 2)
        for k in range(half):
            A.add_chunk(chunks[k]["b"], chunks[k]["n"], chunks[k]["disp"], 128, int(firsts[k]))
        for k in range(half, len(chunks)):
            B.add_chunk(chunks[k]["b"], chunks[k]["n"], chunks[k]["disp"], 128, int(firsts[k]))
        stride = int(firsts[-1]) * 128
        got = np.full((len(q["m"]), stride), -3, np.int32)
        B.search(stride, out=got)
        A.search(stride, out=got)
        ts, ti, _ = A.search_topr(5, N)
        assert np.array_equal(got[:, :N], want)
        # A's device is current now; B gets the first half as well and is searched again (all chunks resident on B)
        for k in range(half):
            B.add_chunk(chunks[k]["b"], chunks[k]["n"], chunks[k]["disp"], 128, int(firsts[k]))
        B.set_option("tail_mode", 1)
        got_b = np.full((len(q["m"]), stride), -3, np.int32)
        B.search(stride, out=got_b)
        A.clear_db()
        assert np.array_equal(got_b[:, :N], want)
        nA = int(firsts[half]) * 128
        for k in range(len(q["m"])):
            os_, oi = port.topr(want[k, :min(N, nA)], 5)
            assert np.array_equal(ts[k], os_) and np.array_equal(ti[k], oi), k
    finally:
        A.close(); B.close()


@pytest.mark.parametrize("threshold", [0, 200, 1000])
def test_score_profile_kernel(searcher, gin, golden, threshold):
    """The reference's second lookup technique (MICsearch.c:257-313, selected by query_length_threshold, MICsearch.c:39-43): queries of
    at least `threshold` rows through the score-profile kernel, the others through the query-profile pipeline -- all golden cases
    (multi-pass 3200-row query, both promotion rungs), resident and streamed in, and a seeded database with gaps and odd lengths."""
    q, pp, chunked = gin
    N = golden["search"]["n_sequences"]
    try:
        searcher.set_option("sp_threshold", threshold)
        for name, c in golden["search"]["cases"].items():
            sc = _run(searcher, q, chunked, 128, matrix(c["matrix"]), c["open"], c["extend"])
            assert np.array_equal(sc[:, :N], load_npy(f"scores_{name}.npy")), name
            if threshold == 0:
                assert searcher.last_kernel_name(len(q["m"]) - 1) == "swimm::sw_sp_kernel(swimm::SpParams)"
        searcher.clear_db()
        searcher.set_option("lazy_upload", 1)
        searcher.set_option("upload_piece_kib", 16)
        searcher.set_queries(q["a"], q["m"], q["disp"], matrix("blosum62"), 10, 2)
        vc = load_chunks(searcher, chunked, 128)
        sc, _ = searcher.search(vc * 128)
        assert np.array_equal(sc[:, :N], load_npy("scores_blosum62_g10_e2.npy"))
        rng = np.random.default_rng(77 + threshold)
        lens = np.sort(rng.integers(0, 700, 3000)).astype(np.uint16)
        codes = rng.integers(0, 24, int(lens.astype(np.int64).sum())).astype(np.int8)
        qs = [rng.integers(0, 24, int(n)).astype(np.int8) for n in (1, 31, 32, 33, 64, 333, 1001)]
        m = np.array([len(x) for x in qs], np.uint16)
        disp = np.concatenate([[0], np.cumsum(m.astype(np.int64))]).astype(np.uint32)
        offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        qs[5][:300] = codes[offs[2500]:offs[2500] + 300]
        a = np.concatenate(qs)
        searcher.clear_db()
        searcher.set_option("lazy_upload", 0)
        searcher.set_queries(a, m, disp, matrix("pam250"), 7, 1)
        searcher.add_sequences(lens, codes, 0)
        got, _ = searcher.search(3072)
        one = port.assemble_single_chunk(lens.astype(np.int64), codes, 128, 5)
        want = port.search_exact(a, m, disp, one["b"], one["n"], one["disp"], matrix("pam250"), 7, 1, 128)
        assert np.array_equal(got[:, :3000], want[:, :3000])
    finally:
        searcher.set_option("sp_threshold", 65536)
        searcher.set_option("lazy_upload", 0)
        searcher.set_option("upload_piece_kib", 98304)
        searcher.clear_db()
