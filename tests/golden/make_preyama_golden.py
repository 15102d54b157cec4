#!/usr/bin/env python3
"""Generate tests/golden/preyama_golden.json from the COMPILED REFERENCE (oracle/_ref/libref.so):
seeded MAF block pairs + (beg, end, radius, v) -> the block the reference's own pre_yama() returns
(rows: src/start/size/strand/srcSize/text, score) or null.  Build container only.
Cases flagged v == 0 embed the two reference defects of SURVEY.md appendix A.6."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import inputs  # noqa: E402
from oracle import mzoracle as mo  # noqa: E402


def blk(b):
    return None if b is None else dict(score=b.score, rows=[dict(src=r.src, start=r.start, size=r.size, strand=r.strand,
                                                                  srcSize=r.srcSize, text=r.text) for r in b.rows])


def main():
    ref = mo.Reference()
    rng = np.random.default_rng(424242)
    cases = []
    shapes = [(2, 2), (1, 2), (2, 1), (3, 3), (4, 2), (2, 4), (1, 1), (3, 2), (5, 5), (2, 3)]
    while len(cases) < 60:
        n1, n2 = shapes[len(cases) % len(shapes)]
        a1, a2, beg, end = inputs.random_block_pair(rng, n1, n2, int(rng.integers(70, 320)))
        if end - beg < 12:
            continue
        v = len(cases) % 2
        R = int(rng.choice([30, 30, 30, 15, 50]))
        try:
            mo.pre_yama(a1, a2, beg, end, R, v)          # skip inputs on which the reference would exit(1)
        except RuntimeError:
            continue
        want = ref.pre_yama(a1, a2, beg, end, R, v)
        cases.append(dict(a1=blk(a1), a2=blk(a2), beg=beg, end=end, radius=R, v=v, out=blk(want)))
    path = os.path.join(HERE, "preyama_golden.json")
    json.dump(cases, open(path, "w"), separators=(",", ":"))
    print(len(cases), "cases ->", path, os.path.getsize(path), "bytes;",
          sum(c["out"] is None for c in cases), "null outputs,", sum(c["v"] == 0 for c in cases), "with v=0")


if __name__ == "__main__":
    main()
