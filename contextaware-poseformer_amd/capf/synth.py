"""Deterministic synthetic checkpoints and inputs (counter-based, NOT torch RNG).

Everything is keyed by (seed, crc32(name)) through numpy's Philox bit generator, so the build
container (where the reference is imported to make tests/golden) and the GPU box (where the
reference does not exist) regenerate bit-identical weights and inputs.

Scales follow PyTorch's defaults closely enough that activations stay O(0.1..1) through ~300
layers (SURVEY.md §8d "Synthetic inputs"):
  conv / linear weights   U(-1/sqrt(fan_in), +1/sqrt(fan_in))      (kaiming-uniform, a=sqrt(5))
  linear biases           U(-1/sqrt(fan_in), +1/sqrt(fan_in))
  LayerNorm               gamma U(0.8,1.2), beta N(0,0.05^2)
  BatchNorm 'default'     gamma 1, beta 0, mean 0, var 1            (PyTorch init)
  BatchNorm 'random'      gamma U(0.5,1.0), beta/mean N(0,0.1^2), var U(0.8,1.2)   (exercises BN folding)
Unlike the reference's DeformableBlock._reset_parameters (pose_dformer.py:103-113), the
sampling_offsets / attention_weights matrices are NOT zeroed: data-dependent offsets make the
deformable sampler test meaningful.
"""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _uniform(seed, name, shape, lo, hi):
    return _rng(seed, name).uniform(lo, hi, size=shape).astype(np.float32)


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(size=shape) * std).astype(np.float32)


def synth_tensor(name, shape, seed, shapes, bn_mode="random"):
    """Value for one state_dict entry.  `shapes` = {name: shape} of the whole checkpoint (needed
    to find a bias's fan_in from its sibling weight)."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    stem = name[: -len(leaf) - 1] if "." in name else ""
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "Spatial_pos_embed":
        return _normal(seed, name, shape, 0.02)
    is_bn = (stem + ".running_mean") in shapes
    is_ln = (not is_bn) and len(shape) == 1 and leaf in ("weight", "bias") and \
        len(shapes.get(stem + ".weight", ())) == 1
    if is_bn:
        if bn_mode == "default":
            return {"weight": np.ones, "running_var": np.ones}.get(leaf, np.zeros)(shape, dtype=np.float32)
        if leaf == "weight":
            return _uniform(seed, name, shape, 0.5, 1.0)
        if leaf == "running_var":
            return _uniform(seed, name, shape, 0.8, 1.2)
        return _normal(seed, name, shape, 0.1)
    if is_ln:
        return _uniform(seed, name, shape, 0.8, 1.2) if leaf == "weight" else _normal(seed, name, shape, 0.05)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return _uniform(seed, name, shape, -b, b)
    if leaf == "bias":
        wshape = shapes[stem + ".weight"]
        b = 1.0 / np.sqrt(int(np.prod(wshape[1:])))
        return _uniform(seed, name, shape, -b, b)
    raise KeyError(f"synth: no rule for {name} {shape}")


def synth_state_dict(shapes, seed=0, bn_mode="random"):
    """{name: torch tensor} for a {name: shape} schema (e.g. taken from model.state_dict())."""
    shapes = {k: tuple(v) for k, v in shapes.items()}
    return {k: torch.from_numpy(synth_tensor(k, s, seed, shapes, bn_mode)) for k, s in shapes.items()}


def load_synthetic(model, seed=0, bn_mode="random"):
    """Fill any module exposing the reference's state_dict schema, strict=True."""
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed, bn_mode)
    model.load_state_dict(sd, strict=True)
    return sd


def synth_inputs(batch, height=256, width=256, seed=0, crop_range=(192, 256), with_gt=False):
    """Synthetic H36M-shaped inputs:
    images  [B,H,W,3] fp32 NHWC ~ N(0,1)          (post mean/std normalisation, datasets/utils.py:47-50)
    k2d     [B,17,2]  ~ U(-1,1)                   (screen-normalised, H36M-Toolbox/transform.py:92-96)
    kcrop   [B,17,2]  pixel coords U(0,cw-1)xU(0,ch-1); crop_range=(192,256) is reference-faithful,
                      (W,H) exercises out-of-range ref for 256x256 / 384x288 inputs
    gt      [B,1,17,3] ~ N(0,0.3^2), joint 0 zeroed (datasets/utils.py:52-53)
    """
    images = _normal(seed, f"images/{batch}x{height}x{width}", (batch, height, width, 3), 1.0)
    k2d = _uniform(seed, f"k2d/{batch}", (batch, 17, 2), -1.0, 1.0)
    cw, ch = crop_range
    kx = _uniform(seed, f"kcrop.x/{batch}/{cw}", (batch, 17, 1), 0.0, cw - 1.0)
    ky = _uniform(seed, f"kcrop.y/{batch}/{ch}", (batch, 17, 1), 0.0, ch - 1.0)
    out = [torch.from_numpy(images), torch.from_numpy(k2d), torch.from_numpy(np.concatenate([kx, ky], -1))]
    if with_gt:
        gt = _normal(seed, f"gt/{batch}", (batch, 1, 17, 3), 0.3)
        gt[:, :, 0] = 0.0
        out.append(torch.from_numpy(gt))
    return tuple(out)


def adversarial_crop_keypoints(batch, seed=0):
    """Crop keypoints that land on / next to pixel boundaries of every map resolution: exact
    integers, the +-1 borders of the normalised range, -0.0, and +-1 ulp neighbours."""
    base = np.array([0.0, -0.0, 95.0, 96.0, 97.0, 191.0, 192.0, 127.0, 128.0, 129.0, 255.0, 256.0,
                     0.5, 1.5, 3.0, 6.0, 12.0, 24.0, 48.0, 47.999996, 48.000004, 190.99998, 383.0], np.float32)
    r = _rng(seed, f"adv/{batch}")
    k = r.choice(base, size=(batch, 17, 2)).astype(np.float32)
    bump = r.integers(-1, 2, size=k.shape)
    k = np.where(bump < 0, np.nextafter(k, np.float32(-1e9)), np.where(bump > 0, np.nextafter(k, np.float32(1e9)), k))
    return torch.from_numpy(k.astype(np.float32))
