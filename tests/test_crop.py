"""N3: per-frame affine crop = get_affine_transform + cv2.warpAffine(INTER_LINEAR) of
ContextPose/mvn/utils/img.py:16-69.  OpenCV is not available in the build container and the reference has no
test for this path, so parity is UNPINNED: the oracle (oracle/crop_oracle.py) restates OpenCV's published
fixed-point algorithm, is checked here against hand-computable cases, and the HIP kernel is checked bit-for-bit
against the oracle."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401  (sys.path setup)


def _oracle():
    import sys, os
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import crop_oracle
    return crop_oracle


# ---- CPU: the oracle on hand-computable cases, the library's host-side affine against the oracle -----------
def test_oracle_affine_is_scale_plus_translate_about_the_centre():
    co = _oracle()
    m = co.get_affine_transform((500.0, 480.0), (1.5, 2.0), (192, 256))
    s = (192 - 1) / (1.5 * 200.0 - 1)            # dst_w - 1 over src_w - 1 (img.py:30-37)
    assert np.allclose(m, [[s, 0, 95.5 - s * 500.0], [0, s, 127.5 - s * 480.0]], rtol=0, atol=1e-4)
    assert np.allclose(m @ [500.0, 480.0, 1.0], [95.5, 127.5], atol=1e-4)       # the centre lands on the crop centre


def test_oracle_warp_integer_translation_is_an_exact_copy_with_zero_border():
    co = _oracle()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
    m = np.array([[1.0, 0.0, -7.0], [0.0, 1.0, 4.0]])            # dst(x, y) = src(x + 7, y - 4)
    out = co.warp_affine_linear_u8(img, m, 30, 20)
    want = np.zeros((20, 30, 3), np.uint8)
    want[4:20, :, :] = img[0:16, 7:37, :]
    assert np.array_equal(out, want)


def test_oracle_warp_half_pixel_shift_averages_neighbours_and_constant_stays_constant():
    co = _oracle()
    img = np.zeros((4, 6, 3), np.uint8)
    img[:, :, 0] = np.arange(6) * 10                                # a ramp along x in channel 0
    img[:, :, 1] = 200
    out = co.warp_affine_linear_u8(img, np.array([[1.0, 0.0, -0.5], [0.0, 1.0, 0.0]]), 5, 4)   # dst(x) = src(x + 0.5)
    assert np.array_equal(out[:, :, 0], np.tile(np.arange(5) * 10 + 5, (4, 1)))
    assert np.all(out[:, :, 1] == 200)
    out2 = co.warp_affine_linear_u8(img, np.array([[2.0, 0.0, -3.3], [0.0, 2.0, -2.1]]), 2, 2)   # src = (dst + t) / 2: inside
    assert np.all(out2[:, :, 1] == 200)                             # weights sum to exactly 2^15


def test_oracle_warp_exact_2x_downscale_is_the_rounded_mean_of_four_pixels():
    """dst = (src - 0.5) / 2, so dst pixel (x, y) samples src at (2x + 0.5, 2y + 0.5): both fractions are 16/32, the four
    15-bit weights are 8192 each and OpenCV's rounding (acc + 2^14) >> 15 gives floor((a + b + c + d + 2) / 4)."""
    co = _oracle()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(12, 16, 3), dtype=np.uint8)
    m = np.array([[0.5, 0.0, -0.25], [0.0, 0.5, -0.25]])
    out = co.warp_affine_linear_u8(img, m, 7, 5)
    i = img.astype(np.int64)
    want = (i[0:10:2, 0:14:2] + i[0:10:2, 1:15:2] + i[1:11:2, 0:14:2] + i[1:11:2, 1:15:2] + 2) // 4
    assert np.array_equal(out, want.astype(np.uint8))


def test_oracle_warp_non_axis_aligned_cases_worked_by_hand():
    """(a) a quarter turn, x' = (Hs - 1) - y, y' = x: every sample point is a pixel centre -> out[y', x'] = img[Hs-1-x', y'].
    (b) a shear x' = x + y / 4: dst row y' samples src at x = x' - y'/4, i.e. fraction f = (-y'/4) mod 1 in {0, 3/4, 1/2, 1/4}
    (all multiples of 1/32), weights ((1-f) 2^15, f 2^15), value ((1-f) a + f b) with OpenCV's (acc + 2^14) >> 15."""
    co = _oracle()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    hs = img.shape[0]
    out = co.warp_affine_linear_u8(img, np.array([[0.0, -1.0, hs - 1.0], [1.0, 0.0, 0.0]]), hs, 11)      # out is [11 rows, 9 cols]
    want = np.zeros((11, hs, 3), np.uint8)
    for yp in range(11):
        for xp in range(hs):
            want[yp, xp] = img[hs - 1 - xp, yp]
    assert np.array_equal(out, want)
    out = co.warp_affine_linear_u8(img, np.array([[1.0, 0.25, 0.0], [0.0, 1.0, 0.0]]), 8, 8)
    i = img.astype(np.int64)
    for yp in range(8):
        for xp in range(3, 8):                      # columns whose two source pixels are inside the image
            xs = xp - yp / 4.0
            x0 = int(np.floor(xs))
            f32 = int(round((xs - x0) * 32))         # exact: multiples of 8
            w1 = f32 * 1024
            w0 = 32768 - w1
            val = (i[yp, x0] * w0 + (i[yp, x0 + 1] * w1 if w1 else 0) + 16384) >> 15
            assert np.array_equal(out[yp, xp], val.astype(np.uint8)), (yp, xp)


@pytest.mark.parametrize("center,scale", [((500.0, 480.0), (1.5, 2.0)), ((512.3, 431.7), (1.037, 1.3826)),
                                          ((100.0, 900.0), (2.25, 3.0))])
def test_library_affine_matches_the_oracle(center, scale):
    from capf import lib as capf
    co = _oracle()
    got = capf.affine_from_center_scale(center, scale, (192, 256))
    want = co.get_affine_transform(center, scale, (192, 256))
    assert got.shape == (2, 3)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-10)


def test_mirror_module_keeps_the_reference_signature():
    from mvn.utils import img
    m = img.get_affine_transform((500.0, 480.0), (1.5, 2.0), 0, (192, 256))
    assert m.shape == (2, 3) and m[0, 1] == 0.0 or abs(m[0, 1]) < 1e-9
    with pytest.raises(ValueError):
        img.get_affine_transform((500.0, 480.0), (1.5, 2.0), 30, (192, 256))


# ---- GPU: the warp kernel against the oracle, bit for bit ---------------------------------------------------
@pytest.mark.gpu
def test_warp_kernel_is_bit_exact_against_the_oracle():
    import torch
    from capf import lib as capf
    co = _oracle()
    rng = np.random.default_rng(7)
    frames, mats, wants = [], [], []
    cases = [((1002, 1000), (512.3, 431.7), (1.037, 1.3826)),      # Human3.6M frame sizes and typical boxes
             ((1000, 1000), (480.0, 520.0), (1.5, 2.0)),
             ((1002, 1000), (60.0, 950.0), (1.9, 2.5333)),          # box hanging over the image border -> zero fill
             ((300, 200), (100.0, 150.0), (0.4, 0.5333)),           # up-sampling crop
             ((64, 48), (-20.0, 10.0), (0.6, 0.8))]                 # mostly outside
    for (h, w), c, s in cases:
        f = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        m = co.get_affine_transform(c, s, (192, 256))
        frames.append(torch.from_numpy(f).cuda())
        mats.append(m)
        wants.append(co.warp_affine_linear_u8(f, m, 192, 256))
    mats.append(np.array([[1.0, 0.0, -7.0], [0.0, 1.0, 4.0]]))      # integer translation: exact copy
    f = rng.integers(0, 256, size=(300, 260, 3), dtype=np.uint8)
    frames.append(torch.from_numpy(f).cuda())
    wants.append(co.warp_affine_linear_u8(f, mats[-1], 192, 256))
    out = capf.warp_affine(frames, np.stack(mats), (192, 256)).cpu().numpy()
    for i, want in enumerate(wants):
        assert np.array_equal(out[i], want), f"frame {i}: {np.abs(out[i].astype(int) - want.astype(int)).max()} max diff"


@pytest.mark.gpu
def test_crop_image_batch_feeds_the_prefetcher():
    """crop (N3) -> data_prefetcher-style preprocessing (N1) on the GPU, shapes and value range only."""
    import torch
    from mvn.utils import img
    rng = np.random.default_rng(3)
    frames = [torch.from_numpy(rng.integers(0, 256, size=(1002, 1000, 3), dtype=np.uint8)).cuda() for _ in range(4)]
    crops = img.crop_image_batch(frames, [(500.0, 500.0)] * 4, [(1.5, 2.0)] * 4, (192, 256))
    assert crops.shape == (4, 256, 192, 3) and crops.dtype == torch.uint8
    one = img.crop_image(frames[1], (500.0, 500.0), (1.5, 2.0), (192, 256))
    assert torch.equal(one, crops[1])
