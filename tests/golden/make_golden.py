#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own CPU path.

Run in the build container only (needs /root/reference to build oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

Inputs are synthetic (swimm_amd.synth, fixed seeds).  Outputs are what the reference code
(oracle/_ref/libswimm_ref.so = CPUsearch.c + sequences.c + utils.c + submat.c compiled
unmodified) produces for them: preprocessed-DB bytes, query layout, interleaved DB layout,
full score vectors for several matrices / gap settings, and the sorted order.  Only data is
written here -- FASTA inputs, integer arrays, hashes -- never reference source text.

Known reference quirk kept OUT of the fixtures: preprocess_db leaves one uninitialised byte
at the end of every title it writes to .desc (sequences.c:112-116); the fixture stores the
clean titles and the test tolerates that single trailing byte on the reference side.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from oracle import ref  # noqa: E402
from swimm_amd import synth  # noqa: E402

SEED = 11


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def build_inputs():
    qs = synth.make_queries(SEED, [144, 189, 375])
    hom = synth.planted_homologs(SEED, qs)
    wrep = np.full(3200, ord("W"), dtype=np.uint8)
    hom.append(("syn|WREP|W x 3200 (self score 11*3200 under BLOSUM62: int32 tier)", wrep))
    L = synth.lengths_normal(SEED, 400, 200, 80, 30, 700)
    db = synth.make_db(SEED, L, planted=hom)
    queries = qs + [("syn|WQ|W x 3200 query", wrep)]
    return db, queries


def main():
    db, queries = build_inputs()
    db_fa = os.path.join(HERE, "db_small.fasta")
    q_fa = os.path.join(HERE, "queries_small.fasta")
    synth.write_fasta(db_fa, synth.db_records(db), width=60)
    synth.write_fasta(q_fa, queries, width=70)

    G = {"seed": SEED, "db_fasta": "db_small.fasta", "query_fasta": "queries_small.fasta"}
    G["submat_sha256"] = {name: sha(ref.submat(name).tobytes()) for name in ref.MATRICES}

    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "db")
        ref.preprocess_db(db_fa, prefix, 2)
        seq = open(prefix + ".seq", "rb").read()
        info = open(prefix + ".info", "rb").read()
        desc = open(prefix + ".desc", "rb").read().split(b"\n")
        n, D, mt = (int(x) for x in info.split())
        titles_sorted = []
        clean = {">" + t for t in db.titles}
        for line in desc[:n]:
            s = line.decode("latin1")
            if s not in clean and s[:-1] in clean:
                s = s[:-1]  # drop the uninitialised trailing byte (see module docstring)
            assert s in clean, s
            titles_sorted.append(s)
        G["preprocess"] = {"info": info.decode(), "seq_sha256": sha(seq), "seq_bytes": len(seq),
                           "desc_clean_sha256": sha(("\n".join(titles_sorted) + "\n").encode("latin1"))}
        np.save(os.path.join(HERE, "db_small_lengths_sorted.npy"), np.frombuffer(seq[:2 * n], dtype="<u2"))

        rq = ref.load_queries(q_fa, 0, 1)
        qclean = {">" + t for t, _ in queries}  # same uninitialised trailing byte as .desc (sequences.c:331-335)
        rq["titles"] = [t if t in qclean else t[:-1] for t in rq["titles"]]
        assert all(t in qclean for t in rq["titles"])
        rq1 = ref.load_queries(q_fa, 1, 1)
        G["queries"] = {
            "mode0": {"a_sha256": sha(rq["a"].tobytes()), "m": rq["m"].tolist(), "lengths": rq["lengths"].tolist(),
                      "disp": rq["disp"].tolist(), "titles": rq["titles"], "Q": rq["Q"]},
            "mode1": {"a_sha256": sha(rq1["a"].tobytes()), "m": rq1["m"].tolist(), "Q": rq1["Q"]},
        }

        asm = {}
        for vl, blk in ((32, 60), (16, 125)):
            rs = ref.assemble_single_chunk(prefix, vl, blk, 1)
            asm[f"single_vl{vl}_b{blk}"] = {"b_sha256": sha(rs["b"].tobytes()), "n": rs["n"].tolist(),
                                             "nbbs": rs["nbbs"].tolist(), "disp": [int(x) for x in rs["disp"]],
                                             "vD": rs["vD"], "max_length": rs["max_length"]}
        for vl, mx in ((16, 20000), (32, 50000)):
            rm = ref.assemble_multiple_chunks(prefix, vl, mx, 1)
            asm[f"multi_vl{vl}_k{mx}"] = {"vc": rm["vc"], "vD": rm["vD"],
                                           "counts": [c["count"] for c in rm["chunks"]],
                                           "chunk_vD": [int(c["vD"]) for c in rm["chunks"]],
                                           "b_sha256": [sha(c["b"].tobytes()) for c in rm["chunks"]],
                                           "disp_sha256": [sha(c["disp"].tobytes()) for c in rm["chunks"]]}
        G["assemble"] = asm

        rs = ref.assemble_single_chunk(prefix, 32, 60, 1)
        N = rs["sequences_count"]
        cases = {}
        for name, sm, go, ge in (("blosum62_g10_e2", "blosum62", 10, 2), ("pam250_g10_e2", "pam250", 10, 2),
                                 ("blosum50_g10_e2", "blosum50", 10, 2), ("blosum45_g5_e1", "blosum45", 5, 1),
                                 ("pam30_g12_e3", "pam30", 12, 3)):
            sc, _ = ref.cpu_search(rq["a"], rq["m"], rq["disp"], rs["b"], rs["n"], rs["nbbs"], rs["disp"],
                                   ref.submat(sm), go, ge, 32, threads=4)
            sc = sc[:, :N].astype(np.int32)
            # SSE path must agree
            rs16 = ref.assemble_single_chunk(prefix, 16, 125, 1)
            sc16, _ = ref.cpu_search(rq["a"], rq["m"], rq["disp"], rs16["b"], rs16["n"], rs16["nbbs"], rs16["disp"],
                                     ref.submat(sm), go, ge, 16, threads=3)
            assert np.array_equal(sc, sc16[:, :N]), name
            np.save(os.path.join(HERE, f"scores_{name}.npy"), sc)
            order = np.stack([ref.sort_scores(sc[q], 1 + q % 3)[1] for q in range(sc.shape[0])]).astype(np.int32)
            np.save(os.path.join(HERE, f"order_{name}.npy"), order)
            cases[name] = {"matrix": sm, "open": go, "extend": ge, "max": sc.max(axis=1).tolist()}
        G["search"] = {"n_sequences": int(N), "cases": cases}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote golden fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
