"""Global experiment config with the reference's key schema (ContextPose/mvn/utils/cfg.py:5-181).

`config` is an attribute-dict; `update_config(path)` overlays a YAML file and, like the reference
(cfg.py:166-174), raises ValueError on a key that does not already exist, so the reference's
experiments/human36m/human36m.yaml loads unchanged.  Only `model.backbone.*` and
`model.poseformer.*` are read by the hot path (conpose.py:14-27, pose_dformer.py:167-172); the rest
of the tree is carried so that train.py-style callers find the keys they expect.
"""
import os

import yaml


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __deepcopy__(self, memo):
        import copy
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


_DEFAULTS = yaml.safe_load("""
title: human36m_vol_softmax_single
kind: human36m
azureroot: ''
logdir: logs
batch_output: false
vis_freq: 1000
vis_n_elements: 10
id: 600
frame: 1
model:
  image_shape: [192, 256]
  init_weights: true
  checkpoint: null
  backbone:
    type: hrnet_32
    num_final_layer_channel: 17
    num_joints: 17
    num_layers: 152
    init_weights: true
    fix_weights: false
    checkpoint: data/pretrained/human36m/pose_hrnet_w32_256x192.pth
    NUM_JOINTS: 17
    PRETRAINED_LAYERS: ['*']
    STEM_INPLANES: 64
    FINAL_CONV_KERNEL: 1
    STAGE2: {NUM_MODULES: 1, NUM_BRANCHES: 2, NUM_BLOCKS: [4, 4], NUM_CHANNELS: [32, 64], BLOCK: BASIC, FUSE_METHOD: SUM}
    STAGE3: {NUM_MODULES: 4, NUM_BRANCHES: 3, NUM_BLOCKS: [4, 4, 4], NUM_CHANNELS: [32, 64, 128], BLOCK: BASIC, FUSE_METHOD: SUM}
    STAGE4: {NUM_MODULES: 3, NUM_BRANCHES: 4, NUM_BLOCKS: [4, 4, 4, 4], NUM_CHANNELS: [32, 64, 128, 256], BLOCK: BASIC, FUSE_METHOD: SUM}
    NUM_LAYERS: 50
    DECONV_WITH_BIAS: false
    NUM_DECONV_LAYERS: 3
    NUM_DECONV_FILTERS: [256, 256, 256]
    NUM_DECONV_KERNELS: [4, 4, 4]
  volume_net:
    volume_aggregation_method: softmax
    use_gt_pelvis: false
    cuboid_size: 2500.0
    volume_size: 64
    volume_multiplier: 1.0
    volume_softmax: true
    use_feature_v2v: true
    att_channels: 51
    temperature: 1500
  poseformer: {base_dim: 32, embed_dim_ratio: 128, depth: 4, levels: 4}
loss:
  criterion: MAE
  mse_smooth_threshold: 0
  grad_clip: 0
  scale_keypoints_3d: 0.1
  use_volumetric_ce_loss: true
  volumetric_ce_loss_weight: 0.01
  use_global_attention_loss: true
  global_attention_loss_weight: 1000000
dataset:
  kind: human36m
  data_format: ''
  transfer_cmu_to_human36m: false
  root: ../H36M-Toolbox/images/
  extra_root: data/human36m/extra
  train_labels_path: data/human36m/extra/human36m-multiview-labels-GTbboxes.npy
  val_labels_path: data/human36m/extra/human36m-multiview-labels-GTbboxes.npy
  train_dataset: multiview_human36m
  val_dataset: human36m
train:
  n_objects_per_epoch: 15000
  n_epochs: 9999
  n_iters_per_epoch: 5000
  batch_size: 3
  optimizer: Adam
  backbone_lr: 0.0001
  backbone_lr_step: [1000]
  backbone_lr_factor: 0.1
  process_features_lr: 0.001
  volume_net_lr: 0.001
  volume_net_lr_decay: 0.99
  volume_net_lr_step: [1000]
  volume_net_lr_factor: 0.5
  with_damaged_actions: true
  undistort_images: true
  scale_bbox: 1.0
  ignore_cameras: []
  crop: true
  erase: false
  shuffle: true
  randomize_n_views: true
  min_n_views: 1
  max_n_views: 1
  num_workers: 8
  limb_length_path: data/human36m/extra/mean_and_std_limb_length.h5
  pred_results_path: data/pretrained/human36m/human36m_alg_10-04-2019/checkpoints/0060/results/train.pkl
val:
  flip_test: true
  batch_size: 6
  with_damaged_actions: true
  undistort_images: true
  scale_bbox: 1.0
  ignore_cameras: []
  crop: true
  erase: false
  shuffle: false
  randomize_n_views: true
  min_n_views: 1
  max_n_views: 1
  num_workers: 10
  retain_every_n_frames_in_test: 1
  limb_length_path: data/human36m/extra/mean_and_std_limb_length.h5
  pred_results_path: data/pretrained/human36m/human36m_alg_10-04-2019/checkpoints/0060/results/val.pkl
""")

config = AttrDict(_DEFAULTS)


def update_dict(overlay, cfg):
    for key, val in overlay.items():
        if key not in cfg:
            raise ValueError("{} not exist in cfg.py".format(key))
        if isinstance(val, dict):
            update_dict(val, cfg[key])
        else:
            cfg[key] = val


def update_config(path):
    with open(path) as fin:
        update_dict(yaml.safe_load(fin), config)


def _prefix_data_paths(node, root):
    for key, val in node.items():
        if isinstance(val, str) and val.startswith("data/"):
            node[key] = os.path.join(root, val)
        elif isinstance(val, dict):
            _prefix_data_paths(val, root)


def update_dir(azureroot, logdir):
    config.azureroot = azureroot
    config.logdir = os.path.join(azureroot, logdir)
    ckpt = config.model.checkpoint
    if ckpt is not None and not ckpt.startswith("data/"):
        config.model.checkpoint = os.path.join(azureroot, ckpt)
    _prefix_data_paths(config, azureroot)


def backbone_preset(cfg, backbone):
    """The per-backbone patch train.py:266-277 applies after parsing --backbone."""
    cfg.model.backbone.type = backbone
    if backbone == "hrnet_32":
        cfg.model.poseformer.base_dim = 32
    elif backbone == "hrnet_48":
        cfg.model.backbone.checkpoint = "data/pretrained/coco/pose_hrnet_w48_256x192.pth"
        cfg.model.backbone.STAGE2.NUM_CHANNELS = [48, 96]
        cfg.model.backbone.STAGE3.NUM_CHANNELS = [48, 96, 192]
        cfg.model.backbone.STAGE4.NUM_CHANNELS = [48, 96, 192, 384]
        cfg.model.poseformer.base_dim = 48
    elif backbone == "cpn":
        cfg.train.batch_size = 256
        cfg.model.backbone.checkpoint = "data/pretrained/coco/CPN50_256x192.pth.tar"
        cfg.model.poseformer.base_dim = 256
    else:
        raise ValueError("unknown backbone {}".format(backbone))
    return cfg
