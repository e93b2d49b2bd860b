"""Golden-vector case table, shared by oracle/make_goldens.py (writer, build container only) and
the tests (readers).  Pure data: inputs and weights are regenerated from these seeds by
capf.synth on both sides; only the reference's OUTPUTS are stored in tests/golden/<name>.npz."""

CASES = {
    # name: backbone, B, H, W, weight seed, input seed, BN mode, crop keypoint range (w,h) or 'adv'
    "w32_256x256_b2": dict(backbone="hrnet_32", B=2, H=256, W=256, wseed=1, iseed=3, bn="random", crop=(256, 256)),
    "w32_256x192_b1": dict(backbone="hrnet_32", B=1, H=256, W=192, wseed=2, iseed=4, bn="random", crop=(192, 256)),
    "w32_256x256_adv": dict(backbone="hrnet_32", B=2, H=256, W=256, wseed=1, iseed=5, bn="default", crop="adv"),
    "w48_256x256_b1": dict(backbone="hrnet_48", B=1, H=256, W=256, wseed=6, iseed=7, bn="random", crop=(256, 256)),
    "cpn_384x288_b1": dict(backbone="cpn", B=1, H=384, W=288, wseed=8, iseed=9, bn="random", crop=(288, 384)),
    # MPI-INF-3DHP variant (ContextPose_mpi): no deformable blocks, embed 64, output [B,3,1,17,1]
    "mpi_w32_e64_b2": dict(backbone="hrnet_32", B=2, H=256, W=192, wseed=12, iseed=13, bn="random", crop=(192, 256), mpi=True),
    # depth != levels (ContextPose_mpi/model/pose_dformer.py:199): two blocks per group over the four feature levels
    "mpi_w32_e64_d2_b2": dict(backbone="hrnet_32", B=2, H=256, W=192, wseed=16, iseed=17, bn="random", crop=(192, 256), mpi=True, depth=2),
    # ... and its shipped default: HRNet-48, embed 96 (run_3dhp.py:219-221, common/cfg.py:81-82); B = 2: the reference`s .squeeze() (:240) breaks at B = 1
    "mpi_w48_e96_b2": dict(backbone="hrnet_48", B=2, H=256, W=192, wseed=14, iseed=15, bn="random", crop=(192, 256), mpi=True),
    "cpn_256x192_b1": dict(backbone="cpn", B=1, H=256, W=192, wseed=8, iseed=10, bn="random", crop=(192, 256)),
}


def case_inputs(case):
    """(images, k2d, kcrop) for a case, via capf.synth (deterministic, torch-RNG free)."""
    from capf import synth
    img, k2d, kc = synth.synth_inputs(case["B"], case["H"], case["W"], seed=case["iseed"],
                                      crop_range=(192, 256) if case["crop"] == "adv" else case["crop"])
    if case["crop"] == "adv":
        kc = synth.adversarial_crop_keypoints(case["B"], seed=case["iseed"])
    return img, k2d, kc


def metric_inputs(n=96, seed=77):
    """Seeded evaluation-shaped data for the N2 metrics: pred / gt [n,1,17,3] fp32 (metres, root-relative-ish),
    action_idx [n] (6 actions in interleaved runs, so masks are NOT contiguous), validity [n,17,1]."""
    import numpy as np
    rng = np.random.Generator(np.random.Philox(key=[seed, 20260928]))
    gt = (rng.standard_normal((n, 1, 17, 3)) * 0.3).astype(np.float32)
    gt[:, :, 0] = 0
    # prediction = rotated, scaled, shifted, noisy ground truth (so Procrustes alignment has something to undo)
    ang = rng.uniform(-0.4, 0.4, size=(n, 3))
    pred = np.empty_like(gt)
    for i in range(n):
        cx, cy, cz = np.cos(ang[i]); sx, sy, sz = np.sin(ang[i])
        R = np.array([[cy * cz, -cy * sz, sy], [sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy],
                      [-cx * sy * cz + sx * sz, cx * sy * sz + sx * cz, cx * cy]])
        pred[i, 0] = (gt[i, 0].astype(np.float64) @ R * rng.uniform(0.8, 1.2) + rng.standard_normal(3) * 0.05
                      + rng.standard_normal((17, 3)) * 0.02).astype(np.float32)
    action_idx = ((np.arange(n) // 5) % 6).astype(np.int32)
    validity = (rng.uniform(size=(n, 17, 1)) > 0.2).astype(np.float32)
    return pred, gt, action_idx, validity


PREFETCH_MODES = {   # name: backbone, is_train, flip_test, flip  (datasets/utils.py:33-82)
    "hrnet_eval": ("hrnet_32", False, False, False),
    "hrnet_train_flip": ("hrnet_32", True, False, True),
    "hrnet_train_noflip": ("hrnet_32", True, False, False),
    "hrnet_fliptest": ("hrnet_48", False, True, False),
    "cpn_eval": ("cpn", False, False, False),
    "cpn_fliptest": ("cpn", False, True, False),
}


def prefetch_inputs(B=3, H=8, W=6, seed=91):
    """One loader batch as Human36M.__getitem__ collates it: uint8 BGR crops [B,H,W,3] (every byte value occurs: the
    /255 path is exercised on all 256 inputs), gt [B,1,17,3], k2d [B,17,2], crop keypoints [B,17,2] in 192x256 pixels."""
    import numpy as np
    import torch
    rng = np.random.Generator(np.random.Philox(key=[seed, 20260928]))
    img = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    img.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    gt = (rng.standard_normal((B, 1, 17, 3)) * 0.3).astype(np.float32)
    k2d = rng.uniform(-1, 1, size=(B, 17, 2)).astype(np.float32)
    kc = (rng.uniform(0, 1, size=(B, 17, 2)) * np.array([191.0, 255.0])).astype(np.float32)
    return [torch.from_numpy(a) for a in (img, gt, k2d, kc)]
