"""Host mirror of the crop helpers of ContextPose/mvn/utils/img.py: same names and
argument meaning, MI355X-native underneath (libcapf.so; no OpenCV, no CPU fallback for the warp)."""
import numpy as np

from capf import lib as _capf

IMAGENET_MEAN, IMAGENET_STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])    # img.py:8


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """img.py:16-48.  The reference only ever calls it with rot = 0, shift = 0, inv = 0 (crop_image, :63)."""
    if rot != 0 or inv != 0 or np.any(np.asarray(shift) != 0):
        raise ValueError("only rot = 0, shift = 0, inv = 0 (the reference's only call site, img.py:63) is supported")
    return _capf.affine_from_center_scale(center, scale, output_size)


def crop_image(image, center, scale, output_size):
    """img.py:51-69 for ONE frame: `image` is a uint8 CUDA tensor [H, W, 3]; returns a uint8 CUDA tensor
    [output_size[1], output_size[0], 3]."""
    return crop_image_batch([image], [center], [scale], output_size)[0]


def crop_image_batch(images, centers, scales, output_size):
    """What Human36M.__getitem__ (human36m.py:298-300) does per sample, for a whole batch in one launch:
    images: list of uint8 CUDA tensors [H_i, W_i, 3]; centers / scales: per-frame (x, y) pairs."""
    mats = np.stack([get_affine_transform(c, s, 0, output_size) for c, s in zip(centers, scales)])
    return _capf.warp_affine(list(images), mats, output_size)


def imread(path_or_bytes, device="cuda"):
    """cv2.imread(path, cv2.IMREAD_COLOR | cv2.IMREAD_IGNORE_ORIENTATION) of Human36M.__getitem__ (datasets/human36m.py:292-295) for a baseline
    JPEG: uint8 CUDA tensor [H, W, 3], BGR.  The host walks the Huffman stream, the GPU does the rest (capf_jpeg_decode); files this path does
    not take (progressive, CMYK, ...) raise CapfError -- there is no CPU fallback in this package."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    return _capf.jpeg_decode(bytes(data), device)


def load_and_crop_batch(paths_or_bytes, centers, scales, output_size, device="cuda"):
    """The whole per-sample image path of Human36M.__getitem__ (human36m.py:292-300) for a batch: decode every frame on the GPU, then ONE
    warp launch for all crops.  Returns uint8 CUDA [B, output_size[1], output_size[0], 3], what the prefetcher (capf_preprocess) consumes."""
    return crop_image_batch([imread(p, device) for p in paths_or_bytes], centers, scales, output_size)
