"""TEST INFRASTRUCTURE (not product code): numpy restatement of what libjpeg's DEFAULT decode path does after entropy decoding -- the
second half of cv2.imread / PIL.Image.open(...).convert for a baseline JPEG (reference call site: ContextPose/mvn/datasets/human36m.py:292-295,
cv2.imread(path, IMREAD_COLOR | IMREAD_IGNORE_ORIENTATION)).  The reference delegates the decode to OpenCV -> libjpeg(-turbo); neither is part
of /root/reference, so the algorithm is restated from libjpeg's published sources (IJG libjpeg 6b / libjpeg-turbo 3.x):
    jidctint.c   jpeg_idct_islow        13-bit fixed point inverse DCT, two passes, range limit (JDCT_ISLOW is the default dct_method)
    jdsample.c   h2v1 / h2v2 fancy      triangle-filter chroma upsampling (do_fancy_upsampling = TRUE is the default)
    jdcolor.c    ycc_rgb_convert        16-bit fixed point YCbCr -> RGB
    jdmainct.c                          context rows: the row above the first and below the last TRUE chroma row are replicas
PINNED: tests/test_jpeg.py checks this restatement bit for bit against Pillow 12.2's bundled libjpeg-turbo (the third-party decoder the
image ships; `PIL.features.version('jpg')`) on every sampling mode, odd sizes, restart intervals and several qualities, and against the
committed fixture tests/golden/jpeg_cases.npz (written by oracle/make_jpeg_goldens.py from that same Pillow).  OpenCV itself is not in the
image: that cv2.imread equals a libjpeg-turbo default decode is libjpeg's API contract, not something this repository can run.

Inputs are the quantised coefficient blocks (natural order) -- e.g. from capf.lib.jpeg_coefficients, the HOST half of the product decoder --
and the quantisation tables; so a test that compares `decode_from_coefficients` with Pillow also pins that host half."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
CB, P1 = 13, 2
F = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069, f2053=16819,
         f2562=20995, f3072=25172)


def _pass(v, shift):
    """jidctint.c: one 1-D pass over the LAST axis of v [..., 8] (int64), descaled by `shift`."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["f0541"]
    tmp2 = z1 + z3 * (-F["f1847"])
    tmp3 = z1 + z2 * F["f0765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0, tmp1 = (z2 + z3) << CB, (z2 - z3) << CB
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["f0298"], tmp1 * F["f2053"], tmp2 * F["f3072"], tmp3 * F["f1501"]
    z1, z2, z3, z4 = z1 * -F["f0899"], z2 * -F["f2562"], z3 * -F["f1961"] + z5, z4 * -F["f0390"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    r = 1 << (shift - 1)
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)
    return (out + r) >> shift


def idct_islow(coef, qt):
    """coef [bh, bw, 64] int16 natural order, qt [64] natural order -> samples [bh * 8, bw * 8] uint8."""
    bh, bw, _ = coef.shape
    v = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(bh, bw, 8, 8)          # [.., row, col]
    ws = _pass(v.swapaxes(-1, -2), CB - P1).swapaxes(-1, -2)                         # pass 1 runs down the columns
    o = _pass(ws, CB + P1 + 3) & 1023                                                # pass 2 along the rows; range_limit[x & RANGE_MASK]
    s = np.where(o < 128, o + 128, np.where(o < 512, 255, np.where(o < 896, 0, o - 896)))
    return s.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8).astype(np.uint8)


def upsample_fancy(pl, dw, dh, hs, vs, W, H):
    """chroma plane pl (padded) with true size dh x dw -> [H, W] int at full resolution (jdsample.c fullsize / h2v1_fancy / h2v2_fancy)."""
    p = pl[:dh, :dw].astype(np.int64)
    if hs == 1 and vs == 1:
        return p[:H, :W]
    if vs == 1:
        prev = np.concatenate([p[:, :1], p[:, :-1]], 1)
        nxt = np.concatenate([p[:, 1:], p[:, -1:]], 1)
        even = (p * 3 + prev + 1) >> 2
        odd = (p * 3 + nxt + 2) >> 2
        even[:, 0], odd[:, -1] = p[:, 0], p[:, -1]
        out = np.empty((dh, 2 * dw), np.int64)
        out[:, 0::2], out[:, 1::2] = even, odd
        return out[:H, :W]
    up = np.concatenate([p[:1], p[:-1]], 0)                                           # context rows: replicas at the top and bottom edges
    dn = np.concatenate([p[1:], p[-1:]], 0)
    rows = np.empty((2 * dh, dw), np.int64)
    rows[0::2], rows[1::2] = p * 3 + up, p * 3 + dn                                   # "colsum" of each output row
    prev = np.concatenate([rows[:, :1], rows[:, :-1]], 1)
    nxt = np.concatenate([rows[:, 1:], rows[:, -1:]], 1)
    even = (rows * 3 + prev + 8) >> 4
    odd = (rows * 3 + nxt + 7) >> 4
    even[:, 0] = (rows[:, 0] * 4 + 8) >> 4
    odd[:, -1] = (rows[:, -1] * 4 + 7) >> 4
    out = np.empty((2 * dh, 2 * dw), np.int64)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:H, :W]


def decode_from_coefficients(coefs, qts, W, H, hs, vs):
    """coefs: per component [bh, bw, 64] int16 natural order; qts: per component [64] natural order -> uint8 BGR [H, W, 3]."""
    Y = idct_islow(coefs[0], qts[0])[:H, :W].astype(np.int64)
    if len(coefs) == 1:
        return np.repeat(Y[..., None], 3, 2).astype(np.uint8)
    dw, dh = -(-W // hs), -(-H // vs)
    cb = upsample_fancy(idct_islow(coefs[1], qts[1]), dw, dh, hs, vs, W, H) - 128
    cr = upsample_fancy(idct_islow(coefs[2], qts[2]), dw, dh, hs, vs, W, H) - 128
    r = Y + ((91881 * cr + 32768) >> 16)
    b = Y + ((116130 * cb + 32768) >> 16)
    g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def pillow_tables(im):
    """PIL's im.quantization -> tables per table index (luma 0, chroma 1) as int64 [64].  Pillow >= 8.3 hands them out in NATURAL (row-major)
    order -- Pillow 12.2, the one in the image, does: the standard luminance table starts 16 11 10 16 24 40 51 61 there."""
    return {k: np.asarray(t, np.int64) for k, t in im.quantization.items()}
