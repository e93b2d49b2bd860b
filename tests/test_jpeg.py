"""N3, first half: the JPEG decode in front of the affine crop -- cv2.imread of Human36M.__getitem__
(ContextPose/mvn/datasets/human36m.py:292-295).  OpenCV is not in the image; cv2.imread and Pillow both decode through libjpeg(-turbo)'s default
path, so the pin is Pillow's bundled libjpeg-turbo: committed fixtures (tests/golden/jpeg_cases.npz, oracle/make_jpeg_goldens.py) and
files encoded on the spot.  Everything is held BIT-EXACT: the host Huffman half + the numpy restatement (oracle/jpeg_oracle.py) on the CPU,
the HIP kernels on the GPU."""
import io
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))

CASES = ["rgb444_q90", "rgb420_q75_odd", "rgb422_q50", "rgb420_q95_opt", "rgb420_q85_rst", "rgb444_q30", "grey_q80", "rgb420_q100_sat"]


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"), allow_pickle=False)


def _quant_tables(data):
    """natural-order quantisation tables per component, parsed from the file's DQT / SOF segments (independent of Pillow)."""
    import jpeg_oracle as jo
    tabs, comp_tq, p = {}, [], 2
    while p < len(data):
        assert data[p] == 0xFF
        m, ln = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        seg = data[p + 4:p + 2 + ln]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                assert seg[q] >> 4 == 0
                t = np.zeros(64, np.int64)
                t[jo.ZIGZAG] = np.frombuffer(bytes(seg[q + 1:q + 65]), np.uint8)
                tabs[seg[q] & 15] = t
                q += 65
        elif m in (0xC0, 0xC1):
            comp_tq = [seg[6 + 3 * i + 2] for i in range(seg[5])]
        elif m == 0xDA:
            break
        p += 2 + ln
    return [tabs[t] for t in comp_tq]


def _oracle_decode(data):
    import jpeg_oracle as jo
    from capf import lib as capf
    info = capf.jpeg_info(data)
    coefs = capf.jpeg_coefficients(data)
    return jo.decode_from_coefficients(coefs, _quant_tables(data), info["width"], info["height"], info["h_samp"], info["v_samp"]), info


@pytest.mark.parametrize("name", CASES)
def test_host_half_and_restatement_reproduce_the_libjpeg_goldens(name):
    """capf_jpeg_info + capf_jpeg_coefficients (the product's host half: marker parse, Huffman decode, DC prediction, restart markers) followed
    by the numpy restatement of libjpeg's islow IDCT / fancy upsampling / colour conversion == what Pillow's libjpeg-turbo decoded, bit for bit."""
    g = _golden()
    data = g[name + ":jpeg"].tobytes()
    got, info = _oracle_decode(data)
    assert [info["width"], info["height"], info["components"], info["h_samp"], info["v_samp"]] == g[name + ":info"].tolist()
    assert np.array_equal(got, g[name + ":bgr"])


def test_restatement_against_the_installed_pillow_on_fresh_files():
    """The same on files encoded on the spot (every sampling mode x several sizes / qualities / restart intervals / optimised tables): the
    pin is the libjpeg-turbo in THIS environment, not only the one that wrote the fixture."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(5)
    n = 0
    for H, W in ((8, 8), (17, 33), (64, 48), (121, 75)):
        y, x = np.mgrid[0:H, 0:W]
        img = np.clip(np.stack([128 + 100 * np.sin(x / 6.0), 128 + 90 * np.cos(y / 5.0 + x / 9.0), (x * 5 + y * 3) % 256], -1)
                      + rng.normal(0, 15, (H, W, 3)), 0, 255).astype(np.uint8)
        for sub in (0, 1, 2):
            for q, kw in ((92, {}), (60, dict(optimize=True)), (80, dict(restart_marker_blocks=1))):
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, "JPEG", quality=q, subsampling=sub, **kw)
                data = buf.getvalue()
                want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1]
                got, _ = _oracle_decode(data)
                assert np.array_equal(got, want), (H, W, sub, q, kw)
                n += 1
    assert n == 36


def test_golden_recipe_regenerates_the_committed_fixture():
    """oracle/make_jpeg_goldens.py on this environment's Pillow reproduces the committed files and decodes (same encoder, same decoder)."""
    pytest.importorskip("PIL")
    import make_jpeg_goldens
    rec, old = make_jpeg_goldens.make(), _golden()
    assert set(rec) == set(old.files)
    for k in rec:
        if k not in ("pillow_version", "libjpeg_turbo"):
            assert np.array_equal(rec[k], old[k]), k


def test_files_outside_the_supported_subset_are_refused_not_misdecoded():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from capf import lib as capf
    from capf.lib import CapfError
    img = (np.arange(32 * 32 * 3) % 251).astype(np.uint8).reshape(32, 32, 3)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=80, progressive=True)
    with pytest.raises(CapfError):
        capf.jpeg_info(buf.getvalue())                                  # SOF2
    buf = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(buf, "JPEG", quality=80)
    with pytest.raises(CapfError):
        capf.jpeg_info(buf.getvalue())                                  # four components
    with pytest.raises(CapfError):
        capf.jpeg_info(b"\x89PNG\r\n\x1a\n" + bytes(64))
    good = _golden()["rgb444_q90:jpeg"].tobytes()
    with pytest.raises(CapfError):
        capf.jpeg_info(good[:40])                                       # truncated inside the tables


def _dht_segments(data):
    """(offset of the table-class byte, counts offset, symbol offset, symbol count) of every Huffman table in the file"""
    out, p = [], 2
    while p + 4 <= len(data):
        m, ln = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        if m == 0xC4:
            q = p + 4
            while q < p + 2 + ln:
                cnt = sum(data[q + 1:q + 17])
                out.append((q, q + 1, q + 17, cnt))
                q += 17 + cnt
        elif m == 0xDA:
            break
        p += 2 + ln
    return out


def test_corrupt_huffman_tables_and_streams_are_refused_or_survived():
    """a DHT whose code-length counts do not form a prefix code used to index past the decoder's 512-entry lookup tables, and a
    DC category above 16 reached shifts by >= 64.  The parser now refuses such tables (what libjpeg's jpeg_make_d_derived_tbl does); a stream
    that is merely garbled decodes to SOMETHING or is refused, without touching memory it does not own."""
    from capf import lib as capf
    from capf.lib import CapfError
    good = bytearray(_golden()["rgb420_q75_odd:jpeg"].tobytes())
    tabs = _dht_segments(good)
    assert len(tabs) == 4
    # oversubscribed: three codes of length 1 (the advisor's example), and 255 codes of length 2
    for length, n in ((1, 3), (2, 255)):
        bad = bytearray(good)
        bad[tabs[0][1] + length - 1] = n
        with pytest.raises(CapfError):
            capf.jpeg_info(bytes(bad))
        with pytest.raises(CapfError):
            capf.jpeg_coefficients(bytes(bad))
    # a DC table that names magnitude category 200, an AC table with a 15-bit coefficient
    for t, sym in ((0, 200), (1, 0x0F)):
        bad = bytearray(good)
        tc_off, _, sym_off, cnt = next(x for x in tabs if (good[x[0]] >> 4) == t)
        assert cnt > 0
        bad[sym_off] = sym
        with pytest.raises(CapfError):
            capf.jpeg_coefficients(bytes(bad))
    # byte flips anywhere in the file: refused or decoded, never a crash (run under the same process: a wild write would take pytest down)
    rng = np.random.default_rng(5)
    refused = 0
    for _ in range(300):
        bad = bytearray(good)
        for pos in rng.integers(2, len(bad), size=int(rng.integers(1, 6))):
            bad[pos] = int(rng.integers(0, 256))
        try:
            capf.jpeg_info(bytes(bad))
            capf.jpeg_coefficients(bytes(bad))
        except CapfError:
            refused += 1
    assert 0 < refused < 300
    assert capf.jpeg_coefficients(bytes(good)) is not None               # (and the intact file still decodes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_decode_is_bit_exact_against_the_libjpeg_goldens(name):
    from capf import lib as capf
    g = _golden()
    out = capf.jpeg_decode(g[name + ":jpeg"].tobytes())
    assert out.dtype.is_floating_point is False and tuple(out.shape) == g[name + ":bgr"].shape
    assert np.array_equal(out.cpu().numpy(), g[name + ":bgr"])


@pytest.mark.gpu
def test_gpu_decode_of_frame_sized_files_and_the_decode_crop_pipeline():
    """A 1000 x 1002 frame (Human3.6M's camera resolution) in 4:2:0 and 4:4:4, bit-exact against Pillow; then the whole per-sample image path
    of Human36M.__getitem__ (human36m.py:292-300): decode -> affine crop, against the crop oracle applied to the libjpeg decode."""
    PIL = pytest.importorskip("PIL")
    import torch
    from PIL import Image
    import crop_oracle
    from capf import lib as capf
    from mvn.utils.img import load_and_crop_batch
    rng = np.random.default_rng(11)
    H, W = 1002, 1000
    y, x = np.mgrid[0:H, 0:W]
    img = np.clip(np.stack([128 + 100 * np.sin(x / 37.0) * np.cos(y / 51.0), 128 + 90 * np.cos(x / 25.0 + y / 19.0), (x + 2 * y) % 256], -1)
                  + rng.normal(0, 6, (H, W, 3)), 0, 255).astype(np.uint8)
    files, decoded = [], []
    for sub in (2, 0):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=88, subsampling=sub)
        data = buf.getvalue()
        want = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1])
        got = capf.jpeg_decode(data)
        assert np.array_equal(got.cpu().numpy(), want)
        files.append(data)
        decoded.append(want)
    centers, scales = [(500.0, 480.0), (430.0, 520.0)], [(1.6, 2.13), (1.2, 1.6)]
    crops = load_and_crop_batch(files, centers, scales, (192, 256)).cpu().numpy()
    for i in range(2):
        m = crop_oracle.get_affine_transform(centers[i], scales[i], (192, 256))
        assert np.array_equal(crops[i], crop_oracle.warp_affine_linear_u8(decoded[i], m, 192, 256))
