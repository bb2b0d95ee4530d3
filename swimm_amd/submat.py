"""Substitution tables in the layout the search path uses (reference: submat.c:4-227, submat.h:4-6).

Source of truth: swimm_amd/data/matrices/<name>.txt -- the public NCBI BLOSUM/PAM integer
matrices, stored as a lower triangle in NCBI residue order.  table(name) expands one into the
24 x 32 int8 array indexed [query_code * 32 + db_code] with codes
A0 B1 C2 D3 E4 F5 G6 H7 I8 K9 L10 M11 N12 P13 Q14 R15 S16 T17 V18 W19 X20 Y21 Z22, 23 = J/O/U dummy;
row 23 and columns 23..31 are zero (so the lane-padding code 24 scores 0 against everything).
tools/gen_submat.py emits the same bytes as C for the host library.
"""
from __future__ import annotations

import os

import numpy as np

NAMES = ("blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250")
CODE_LETTERS = "ABCDEFGHIKLMNPQRSTVWXYZ"   # code i <-> CODE_LETTERS[i]
ROWS, COLS = 24, 32
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "matrices")


def parse_triangle(path: str):
    """-> (letters, {(a, b): score}) from a lower-triangle text file."""
    letters, tri = [], {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            tok = line.split()
            a, vals = tok[0], [int(x) for x in tok[1:]]
            letters.append(a)
            if len(vals) != len(letters):
                raise ValueError(f"{path}: row {a} has {len(vals)} values, expected {len(letters)}")
            for b, v in zip(letters, vals):
                tri[(a, b)] = tri[(b, a)] = v
    return letters, tri


def table(name: str) -> np.ndarray:
    name = name.lower()
    if name not in NAMES:
        raise KeyError(f"unknown substitution matrix {name!r}; supported: {', '.join(NAMES)}")
    letters, tri = parse_triangle(os.path.join(_DIR, name + ".txt"))
    if sorted(letters) != sorted(CODE_LETTERS):
        raise ValueError(f"{name}: residue set {letters} != {CODE_LETTERS}")
    t = np.zeros((ROWS, COLS), dtype=np.int8)
    for i, a in enumerate(CODE_LETTERS):
        for j, b in enumerate(CODE_LETTERS):
            t[i, j] = tri[(a, b)]
    return t.reshape(-1)
