"""The `swimm` program: flag surface, file formats and report text (reference: swimm.c, arguments.c).
Mode 0 (explicit host-CPU search) runs here; mode 1 (MI355X) is exercised by the gpu-marked test."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_npy
from swimm_amd import host, submat

SWIMM = os.path.join(ROOT, "swimm_amd", "bin", "swimm")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def run(*args, check=True):
    p = subprocess.run([SWIMM, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if check:
        assert p.returncode == 0, p.stdout + p.stderr
    return p


@pytest.fixture(scope="module")
def dbprefix(tmp_path_factory, golden):
    prefix = str(tmp_path_factory.mktemp("cli") / "db")
    p = run("-S", "preprocess", "-i", os.path.join(GOLDEN, golden["db_fasta"]), "-o", prefix)
    assert "Database size:\t\t\t413 sequences (84979 residues) \n" in p.stdout
    return prefix


def test_preprocess_files(dbprefix, golden):
    g = golden["preprocess"]
    assert sha(open(dbprefix + ".seq", "rb").read()) == g["seq_sha256"]
    assert open(dbprefix + ".info").read() == g["info"]
    assert sha(open(dbprefix + ".desc", "rb").read()) == g["desc_clean_sha256"]


def parse_report(text):
    """-> list of (query title, length, [(score, title)])"""
    out = []
    for blk in re.split(r"\nQuery no\.\t\t\t\d+\n", text)[1:]:
        desc = re.search(r"Query description: \t\t(.*)\n", blk).group(1)
        length = int(re.search(r"Query length:\t\t\t(\d+) residues\n", blk).group(1))
        body = blk.split("\nScore\tSequence description\n")[1].split("\nSearch date:")[0]
        hits = [(int(l.split("\t", 1)[0]), l.split("\t", 1)[1]) for l in body.split("\n") if "\t" in l]
        out.append((desc, length, hits))
    return out


def check_listing(text, golden, dbprefix, case, r):
    sc, order = load_npy(f"scores_{case}.npy"), load_npy(f"order_{case}.npy")
    titles = [l[1:] for l in open(dbprefix + ".desc").read().split("\n")]
    rep = parse_report(text)
    gq = golden["queries"]["mode0"]
    assert [t for t, _, _ in rep] == [t[1:] for t in gq["titles"]]
    assert [L for _, L, _ in rep] == gq["lengths"]          # unpadded lengths are printed
    for qi, (_, _, hits) in enumerate(rep):
        assert len(hits) == r
        assert [s for s, _ in hits] == sc[qi][order[qi][:r]].tolist()
        assert [t for _, t in hits] == [titles[i] for i in order[qi][:r]]   # includes the tie order


def test_search_mode0_full_listing(dbprefix, golden):
    """every row of the reference's sorted listing (-r N), BLOSUM62 and PAM250, both lane widths"""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "0", "-v", "32", "-c", "4", "-r", "413")
    check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 413)
    assert "Substitution matrix:\t\tBLOSUM62\n" in p.stdout and "Gap open penalty:\t\t10\n" in p.stdout
    assert re.search(r"Search speed:\t\t\t\d+\.\d\d GCUPS\n", p.stdout)
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "0", "-v", "16", "-b", "35", "-c", "3", "-s", "pam250", "-r", "1000")
    check_listing(p.stdout, golden, dbprefix, "pam250_g10_e2", 413)   # top = min(N, r), swimm.c:51
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "0", "-s", "pam30", "-g", "12", "-e", "3")
    check_listing(p.stdout, golden, dbprefix, "pam30_g12_e3", 10)      # default -r 10, -v 16


def test_cpu_search_library_all_cases(dbprefix, golden):
    db = host.db_load(dbprefix)
    q = host.queries_load(os.path.join(GOLDEN, golden["query_fasta"]), True)
    N = golden["search"]["n_sequences"]
    for vl, blk in ((32, 60), (16, 125), (128, 20)):
        one = host.assemble_single_chunk(db["lengths"], db["codes"], vl, blk)
        for name, c in golden["search"]["cases"].items():
            sc, wt = host.cpu_search(q["a"], q["m"], q["disp"], one["b"], one["n"], one["disp"], submat.table(c["matrix"]),
                                     c["open"], c["extend"], vl, threads=4, block_size=blk)
            assert np.array_equal(sc[:, :N], load_npy(f"scores_{name}.npy")), (vl, name)


def test_flag_validation(dbprefix, golden):
    q = os.path.join(GOLDEN, golden["query_fasta"])
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-m", "3", check=False).returncode == 1
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-s", "blosum99", check=False).returncode == 1
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-g", "100", "-e", "100", check=False).returncode == 1
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-v", "64", "-m", "0", check=False).returncode == 1
    assert run("-S", "frobnicate", check=False).returncode == 1
    assert run("-S", "search", "-q", q, check=False).returncode == 1
    p = run("-S", "search", "-q", q, "-d", dbprefix + "_missing", "-m", "0", check=False)
    assert p.returncode == 2 and "SWIMM: An error occurred while opening info file." in p.stdout
    p = run("-S", "preprocess", "-i", "/nonexistent.fa", "-o", "/tmp/x", check=False)
    assert p.returncode == 2


@pytest.mark.parametrize("mode", ["1", "2"])
def test_gpu_mode_fails_loudly_without_backend(dbprefix, golden, tmp_path, mode):
    """modes 1 and 2 with no usable back-end are an error, never a CPU fallback"""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    env = dict(os.environ, SWIMM_HIP_LIB=str(tmp_path / "nope.so"))
    p = subprocess.run([SWIMM, "-S", "search", "-q", q, "-d", dbprefix, "-m", mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 5 and "cannot load the MI355X back-end" in p.stdout
    assert "Query no." not in p.stdout


@pytest.mark.gpu
def test_search_mode1_gpu_listing(dbprefix, golden):
    q = os.path.join(GOLDEN, golden["query_fasta"])
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-x", "1", "-r", "30", "-k", "30000")
    check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 30)
    assert "Execution mode:\t\t\tMI355X only (1 GPUs)\n" in p.stdout and "Promoted to int32:\t\t1 alignments\n" in p.stdout
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-r", "413", "-s", "blosum50")   # r > 64: host selection path
    check_listing(p.stdout, golden, dbprefix, "blosum50_g10_e2", 413)


@pytest.mark.gpu
@pytest.mark.parametrize("host_share", ["128", "256", "0"])
def test_search_mode2_hybrid_listing(dbprefix, golden, host_share):
    """mode 2: the shortest sequences on the host CPU, the rest on the GPU, one merged listing (HETsearch.c)"""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    env = dict(os.environ, SWIMM_HYBRID_CPU_SEQUENCES=host_share)
    p = subprocess.run([SWIMM, "-S", "search", "-q", q, "-d", dbprefix, "-m", "2", "-c", "4", "-r", "413"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 413)
    assert "Execution mode:\t\t\tConcurrent host CPU and MI355X (4 CPU threads and 1 GPUs)\n" in p.stdout
    assert f"Host CPU share:\t\t\t{host_share} sequences" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_search_over_several_devices(dbprefix, golden, mode):
    """-x 3 on the one-GPU box through the virtual-device hook: chunks dealt to three contexts, lists merged"""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    env = dict(os.environ, SWIMM_HIP_VIRTUAL_GPUS="3", SWIMM_HYBRID_CPU_SEQUENCES="128")
    p = subprocess.run([SWIMM, "-S", "search", "-q", q, "-d", dbprefix, "-m", mode, "-x", "3", "-k", "12000", "-r", "413", "-c", "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 413)
    assert "(3 GPUs)" in p.stdout or "and 3 GPUs)" in p.stdout
    p = subprocess.run([SWIMM, "-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-x", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 5 and "4 GPUs requested, 3 visible" in p.stdout


@pytest.mark.gpu
def test_profile_flags_select_the_lookup_technique(dbprefix, golden):
    """-p Q|S|A and -u (arguments.c:109-121, swimm.c:81-85): `-p S` aligns every query with the SCORE-profile kernel (the table
    [query residue][column][lane] of MICsearch.c:257-313, built per chunk in LDS), `-p Q` with the query profile, `-p A` -- the
    default -- resolves to the query profile on gfx950.  Every setting must give the reference listing."""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    for flags, technique in ((("-p", "S", "-u", "100"), "Score Profile in LDS"), (("-p", "S"), "Score Profile in LDS"), (("-p", "Q"), "Query Profile in LDS"),
                             (("-p", "A"), "Adaptive Profile (threshold = 567;"), (("-p", "A", "-u", "0"), "Adaptive Profile (threshold = 0;"),
                             (("-u", "65535"), "Adaptive Profile (threshold = 65535;")):
        p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-r", "40", *flags)
        check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 40)
        assert "Profile technique:\t\t" + technique in p.stdout, p.stdout[-700:]
    p = run("-S", "search", "-q", q, "-d", dbprefix, "-m", "2", "-c", "2", "-r", "413", "-p", "S")       # the hybrid queue's GPU workers too
    check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 413)
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-p", "X", check=False).returncode == 1
    assert run("-S", "search", "-q", q, "-d", dbprefix, "-m", "1", "-u", "70000", check=False).returncode == 1


HYBRID_RE = r"Host CPU share:\t\t\t(\d+) sequences \(([\d.]+) seconds\), MI355X (\d+) sequences \(([\d.]+) seconds\)"


@pytest.fixture(scope="module")
def fake_backend(tmp_path_factory):
    """tests/data/fake_backend.c: the C-ABI answered on the host at an injected rate (test infrastructure)"""
    so = str(tmp_path_factory.mktemp("fake") / "libfake_hip.so")
    lib = os.path.join(ROOT, "swimm_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "swimm_amd", "csrc", "host"), os.path.join(ROOT, "tests", "data", "fake_backend.c"),
                           "-o", so, "-L" + lib, "-lswimm_host", "-Wl,-rpath," + lib, "-lpthread"])
    return so


@pytest.fixture(scope="module")
def queue_db(tmp_path_factory):
    """30 000 sequences / 4e6 residues and two queries: enough cells for a search of a second or so on a few host threads"""
    from swimm_amd import synth
    d = tmp_path_factory.mktemp("queue")
    qs = synth.make_queries(12, [222, 464])
    lens = synth.lengths_lognormal(12, 30_000, 130.0, 0.5, 20, 1500)
    db = synth.make_db(12, lens, planted=synth.planted_homologs(12, qs), with_titles=True)
    fa, qfa, prefix = str(d / "db.fa"), str(d / "q.fa"), str(d / "db")
    synth.write_fasta(fa, synth.db_records(db))
    synth.write_fasta(qfa, qs)
    run("-S", "preprocess", "-i", fa, "-o", prefix)
    return prefix, qfa, db.n


def hits_of(text):
    return [(t, L, hits) for t, L, hits in parse_report(text)]


@pytest.mark.parametrize("gcups,create_ms", [(1.0, 0), (6.0, 150), (0.3, 20)])
def test_mode2_queue_balances_any_rates(fake_backend, queue_db, gcups, create_ms):
    """Mode 2 = ONE queue with the host at its short end and the devices at its long end (HETsearch.c:57,96-104).  A
    stand-in device of any speed and start-up delay (tests/data/fake_backend.c): the listing equals mode 0's, every
    sequence is searched exactly once, and both legs end within 25 % of each other -- no probe, no split fixed in advance."""
    prefix, qfa, n = queue_db
    ref = run("-S", "search", "-q", qfa, "-d", prefix, "-m", "0", "-c", "3", "-r", "30")
    env = dict(os.environ, SWIMM_HIP_LIB=fake_backend, FAKE_GPU_GCUPS=str(gcups), FAKE_GPU_CREATE_MS=str(create_ms), SWIMM_HYBRID_MIN_SLAB="20000")
    p = subprocess.run([SWIMM, "-S", "search", "-q", qfa, "-d", prefix, "-m", "2", "-c", "3", "-r", "30", "-k", "400000"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert hits_of(p.stdout) == hits_of(ref.stdout)
    m = re.search(HYBRID_RE, p.stdout)
    assert m, p.stdout[-600:]
    n_cpu, t_cpu, n_gpu, t_gpu = int(m.group(1)), float(m.group(2)), int(m.group(3)), float(m.group(4))
    assert n_cpu + n_gpu == n and n_cpu % 128 == 0 and n_cpu >= 128 and n_gpu >= 128
    assert abs(t_cpu - t_gpu) <= 0.25 * max(t_cpu, t_gpu), (n_cpu, t_cpu, n_gpu, t_gpu)
    blocks, slabs = map(int, re.search(r"Work queue:\t\t\t(\d+) blocks in CPU and (\d+) slabs in MI355X", p.stdout).groups())
    assert blocks >= 2 and slabs >= 2


def test_mode2_queue_edges(fake_backend, dbprefix, golden):
    """the queue on a database of four lane groups: fixed host shares (test hook), several devices, a device that never
    gets a slab because the host was done first; every listing is the reference's"""
    q = os.path.join(GOLDEN, golden["query_fasta"])
    for extra_env, args in (({"SWIMM_HYBRID_CPU_SEQUENCES": "128"}, ()), ({"SWIMM_HYBRID_CPU_SEQUENCES": "0"}, ()),
                            ({"FAKE_GPU_COUNT": "3"}, ("-x", "3", "-k", "9000")), ({"FAKE_GPU_CREATE_MS": "400"}, ()),
                            ({"SWIMM_HYBRID_MIN_SLAB": "1"}, ("-k", "1"))):
        env = dict(os.environ, SWIMM_HIP_LIB=fake_backend, **extra_env)
        p = subprocess.run([SWIMM, "-S", "search", "-q", q, "-d", dbprefix, "-m", "2", "-c", "2", "-r", "413", *args], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        check_listing(p.stdout, golden, dbprefix, "blosum62_g10_e2", 413)
        m = re.search(HYBRID_RE, p.stdout)
        assert m and int(m.group(1)) + int(m.group(3)) == 413
        if "SWIMM_HYBRID_CPU_SEQUENCES" in extra_env:
            assert m.group(1) == extra_env["SWIMM_HYBRID_CPU_SEQUENCES"]


@pytest.mark.gpu
def test_search_mode2_auto_split(tmp_path):
    """mode 2 on the real device, no test hook: host and GPU pull from one queue (HETsearch.c:57,96-104).  The listing
    equals mode 1's and both legs end within 25 % of each other."""
    from swimm_amd import synth
    qs = synth.make_queries(11, [375, 729, 1500])
    lens = synth.lengths_lognormal(11, 400_000, 300.0, 0.55, 30, 4000)
    db = synth.make_db(11, lens, planted=synth.planted_homologs(11, qs), with_titles=True)
    fa, qfa, prefix = str(tmp_path / "db.fa"), str(tmp_path / "q.fa"), str(tmp_path / "db")
    synth.write_fasta(fa, synth.db_records(db))
    synth.write_fasta(qfa, qs)
    run("-S", "preprocess", "-i", fa, "-o", prefix)
    threads = str(min(64, len(os.sched_getaffinity(0))))
    gpu = run("-S", "search", "-q", qfa, "-d", prefix, "-m", "1", "-r", "25")
    auto = run("-S", "search", "-q", qfa, "-d", prefix, "-m", "2", "-c", threads, "-r", "25")
    assert hits_of(auto.stdout) == hits_of(gpu.stdout)
    m = re.search(HYBRID_RE, auto.stdout)
    assert m, auto.stdout[-600:]
    n_cpu, t_cpu, n_gpu, t_gpu = int(m.group(1)), float(m.group(2)), int(m.group(3)), float(m.group(4))
    assert n_cpu >= 128 and n_cpu % 128 == 0 and n_cpu + n_gpu == db.n and n_gpu >= 128
    assert abs(t_cpu - t_gpu) <= 0.25 * max(t_cpu, t_gpu), (n_cpu, t_cpu, n_gpu, t_gpu)
