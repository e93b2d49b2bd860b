"""CPU restatement of the reference's per-frame affine crop — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(capf_affine_from_center_scale / capf_warp_affine in libcapf.so) never does.

What it restates
  * get_affine_transform      ContextPose/mvn/utils/img.py:16-48   (rot = 0, shift = 0: scale + translate)
  * crop_image                ContextPose/mvn/utils/img.py:51-69   = cv2.warpAffine(image, trans, (192, 256),
                                                                     flags=cv2.INTER_LINEAR), constant border 0
    as called by Human36M.__getitem__, ContextPose/mvn/datasets/human36m.py:281-302.

PARITY UNPINNED: the arithmetic lives in OpenCV (`opencv-python` 4.5.5.64, ContextPose/requirements.txt), which is
neither under /root/reference nor installed here, and the reference holds no test or golden vector for this path.
The functions below restate OpenCV's PUBLISHED algorithm for an 8-bit 3-channel INTER_LINEAR warp
(imgproc/src/imgwarp.cpp: cv::getAffineTransform, cv::invertAffineTransform as inlined in cv::warpAffine,
WarpAffineInvoker's 10-bit fixed-point coordinates, remapBilinear's 5-bit fractions and 15-bit weights):
    D = 1 / (M00 M11 - M01 M10);  inverse map (A, b) in double
    adelta[x] = rint(A00 * x * 1024), bdelta[x] = rint(A10 * x * 1024)
    X0 = rint((A01 * y + b0) * 1024) + 16,  Y0 = rint((A11 * y + b1) * 1024) + 16        (rint: half to even)
    X = (X0 + adelta[x]) >> 5,  sx = X >> 5,  ax = X & 31   (same for Y)
    w = (32 - ay)(32 - ax), (32 - ay) ax, ay (32 - ax), ay ax   (x 32 = the 15-bit table entries)
    out = (sum_i w_i * 32 * p_i + 16384) >> 15   with p_i = 0 outside the image
(The table entry for ax = ay = 0 saturates to 32767 in OpenCV; for 8-bit pixels the result is the same pixel.)
"""
import numpy as np


def get_affine_transform(center, scale, output_size):
    """img.py:16-48 with rot = 0, shift = 0, inv = 0 -> 2x3 float64 matrix mapping source pixels to crop pixels."""
    center = np.array(center)
    scale = np.array(scale)
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    src_dir = np.array([0, (src_w - 1) * -0.5], np.float32)
    dst_dir = np.array([0, (dst_w - 1) * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + src_dir
    dst[0, :] = [(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]
    dst[1, :] = np.array([(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]) + dst_dir

    def third(a, b):                      # img.py:11-13
        d = a - b
        return b + np.array([-d[1], d[0]], dtype=np.float32)

    src[2, :] = third(src[0, :], src[1, :])
    dst[2, :] = third(dst[0, :], dst[1, :])
    return affine_from_points(src, dst)


def affine_from_points(src, dst):
    """cv::getAffineTransform: the 2x3 double matrix with M @ [x, y, 1] = (u, v) for three point pairs."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i] = dst[i, 0]
        b[i + 3] = dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def warp_affine_linear_u8(image, m, out_w, out_h):
    """cv2.warpAffine(image, m, (out_w, out_h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for a
    uint8 [H, W, C] image (see the module docstring for the fixed-point pipeline)."""
    h, w = image.shape[:2]
    m = np.asarray(m, np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a00, a01, a10, a11_ = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    b0 = -a00 * m[0, 2] - a01 * m[1, 2]
    b1 = -a10 * m[0, 2] - a11_ * m[1, 2]
    xs = np.arange(out_w, dtype=np.float64)
    ys = np.arange(out_h, dtype=np.float64)
    adelta = np.rint(a00 * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(a10 * xs * 1024.0).astype(np.int64)
    x0 = np.rint((a01 * ys + b0) * 1024.0).astype(np.int64) + 16
    y0 = np.rint((a11_ * ys + b1) * 1024.0).astype(np.int64) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = X & 31, Y & 31
    img = image.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0)

    w00 = ((32 - ay) * (32 - ax) * 32)[..., None]
    w01 = ((32 - ay) * ax * 32)[..., None]
    w10 = (ay * (32 - ax) * 32)[..., None]
    w11 = (ay * ax * 32)[..., None]
    acc = w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1)
    return ((acc + 16384) >> 15).astype(np.uint8)


def crop_image(image, center, scale, output_size):
    """img.py:51-69."""
    return warp_affine_linear_u8(image, get_affine_transform(center, scale, output_size), output_size[0], output_size[1])
