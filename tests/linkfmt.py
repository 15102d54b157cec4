"""The link-image formats of include/mz_amd.h (mz_link_*), restated in numpy for the CPU tests: decode an image into the pools a
kernel would see, encode a result image from the oracle's traceback.  Test infrastructure: the gloo tests of the link exchange put
`oracle_compute` where the product runs mz_link_plan / mz_link_finish on a GPU (tests/test_shard_gloo.py); with a GPU the product's own
result image is compared with this encoder's byte for byte (tests/test_shard_gpu.py)."""
import numpy as np

from multiz_amd import api
from oracle import mzoracle as mo

CLASS_LETTER = np.frombuffer(b"ACGT-N" + b"N" * 10, dtype=np.uint8)         # k_unnib: a canonical letter per class


def al256(x):
    return (int(x) + 255) & ~255


def decode_image(desc, image, exc):
    """-> the batch the image expands to: K, L, M, N, offA, offB, offBand, poolA, poolB (canonical letters), poolLB, poolRB"""
    desc = np.asarray(desc, dtype=np.int64)
    n, colsA, colsB, band = (int(desc[i]) for i in (0, 3, 4, 5))
    at = api.link_parts(desc)
    assert at["bytes"] == image.size == desc[1] and exc.size == desc[2]
    i32 = lambda k: image[at[k]: at[k] + 4 * n].view(np.int32).copy()  # noqa: E731
    i64 = lambda k: image[at[k]: at[k] + 8 * n].view(np.int64).copy()  # noqa: E731
    out = {k: i32(k) for k in ("K", "L", "M", "N")}
    out.update({k: i64(k) for k in ("offA", "offB", "offBand")})

    def letters(start, count):
        nib = image[start: start + count // 2]
        cls = np.empty(count, dtype=np.uint8)
        cls[0::2], cls[1::2] = nib & 15, nib >> 4
        return CLASS_LETTER[cls]
    out["poolA"], out["poolB"] = letters(at["nibA"], colsA), letters(at["nibB"], colsB)
    ln, lb0, rb0, offC, fmt = i32("len"), i32("lb0"), i32("rb0"), i64("offC"), image[at["fmt"]: at["fmt"] + n]
    LB, RB = np.zeros(band, np.int32), np.zeros(band, np.int32)
    for p in range(n):
        M, o, c = int(ln[p]) - 1, int(out["offBand"][p]), int(offC[p])
        if fmt[p] == 2:
            s = image[at["steps"] + c: at["steps"] + c + M]
            LB[o], RB[o] = lb0[p], rb0[p]
            LB[o + 1: o + M + 1] = lb0[p] + np.cumsum(s & 15)
            RB[o + 1: o + M + 1] = rb0[p] + np.cumsum(s >> 4)
        elif fmt[p] == 1:
            base = exc[c: c + 8].view(np.int32)
            LB[o], RB[o] = base[0], base[1]
            LB[o + 1: o + M + 1] = base[0] + np.cumsum(exc[c + 8: c + 8 + M].astype(np.int64))
            RB[o + 1: o + M + 1] = base[1] + np.cumsum(exc[c + 8 + M: c + 8 + 2 * M].astype(np.int64))
        else:
            raw = exc[c: c + 8 * (M + 1)].view(np.int32)
            LB[o: o + M + 1], RB[o: o + M + 1] = raw[: M + 1], raw[M + 1:]
    out["poolLB"], out["poolRB"] = LB, RB
    return out


def ops_in_column_order(M, N, LB, RB, tb, final):
    """the edit script of the reference's traceback (mz_yama.c:257-291) from the oracle's traceback bytes: 0 = C, 1 = I, 2 = D"""
    w = np.asarray(RB, np.int64) - np.asarray(LB, np.int64) + 1
    rowoff = np.concatenate([[0], np.cumsum(w)])
    Cs, Ds, Is = (int(x) for x in final)
    node = 0 if (Cs >= Ds and Cs >= Is) else (2 if Ds >= Is else 1)
    r, c, ops = M, N, []
    while r > 0 or c > 0:
        st = int(tb[rowoff[r] + c - LB[r]])
        ops.append(node)
        if node == 1:
            c, node = c - 1, (st >> 4) & 3
        elif node == 2:
            r, node = r - 1, (st >> 2) & 3
        else:
            r, c, node = r - 1, c - 1, st & 3
    return ops[::-1]


def encode_results(recs):
    """recs: per pair dict(status, badrow, om, f (3 ints), cells, ops (column order)) -> the result image (uint8)"""
    n = len(recs)
    scripts_at = 64 + al256(api.RES_DT.itemsize * n)
    sizes = [((len(r["ops"]) + 3) // 4 + 3) & ~3 for r in recs]
    img = np.zeros(scripts_at + sum(sizes) + 64, dtype=np.uint8)
    rec = img[64: 64 + api.RES_DT.itemsize * n].view(api.RES_DT)
    off = 0
    for p, r in enumerate(recs):
        rec[p] = (r["status"], r["badrow"], r["om"], tuple(r["f"]), off, r["cells"])
        ops = np.asarray(r["ops"] + [0] * (-len(r["ops"]) % 4), dtype=np.uint8).reshape(-1, 4)
        if len(ops):
            img[scripts_at + off: scripts_at + off + len(ops)] = ops[:, 0] | ops[:, 1] << 2 | ops[:, 2] << 4 | ops[:, 3] << 6
        off += sizes[p]
    return img


def oracle_result_image(desc, image, exc):
    """what a rank sends back for this image, computed by the oracle on the image's expansion; (result image, cells, failed)"""
    b = decode_image(desc, image, exc)
    recs, cells, failed = [], 0, 0
    for p in range(len(b["K"])):
        K, L, M, N = (int(b[k][p]) for k in ("K", "L", "M", "N"))
        a0, b0, d0 = int(b["offA"][p]), int(b["offB"][p]), int(b["offBand"][p])
        LB, RB = b["poolLB"][d0: d0 + M + 1], b["poolRB"][d0: d0 + M + 1]
        r = mo.yama(b["poolA"][a0: a0 + K * M].reshape(M, K), b["poolB"][b0: b0 + L * N].reshape(N, L), LB, RB, variant="profile", want_tb=True)
        if r.rc:
            failed += 1
            recs.append(dict(status=r.rc, badrow=mo.check(M, N, LB, RB)[2], om=0, f=(0, 0, 0), cells=0, ops=[]))
            continue
        c = mo.band_cells(LB, RB)
        cells += c
        recs.append(dict(status=0, badrow=-1, om=r.OM, f=r.final, cells=c, ops=ops_in_column_order(M, N, LB, RB, r.tb, r.final)))
    return encode_results(recs), cells, failed


def oracle_compute(sh):
    """stands in for multiz_amd.shard.link_compute on a box without a GPU: same contract -- gives the share (api.Shard) its result
    image and returns dict(cells, failed)"""
    image, exc = sh.host_image()
    res, cells, failed = oracle_result_image(np.asarray(sh.desc), image, exc)
    sh.set_result(res)
    return dict(cells=cells, failed=failed)
