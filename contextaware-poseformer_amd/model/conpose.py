"""VolumetricTriangulationNet — drop-in for ContextPose_mpi/model/conpose.py:15-46 on libcapf.so.

Same call as CA_PF, but the lifter has no DeformableBlocks (ContextPose_mpi/model/pose_dformer.py:234-261),
and forward returns `(x, None)` with x laid out [B, 3, 1, 17, 1] (the `.view(b,1,p,-1,1).permute(0,3,1,2,4)`
at pose_dformer.py:260).  Everything runs through the same C ABI with `context_blocks = 0`."""
from mvn.models.conpose import CA_PF


class VolumetricTriangulationNet(CA_PF):
    def __init__(self, config, device="cuda:0", compute_dtype="fp32"):
        # ContextPose_mpi/common/cfg.py has no backbone.type key: the width list names the backbone
        width = config.model.backbone.STAGE2.NUM_CHANNELS[0]
        if width not in (32, 48):
            raise NotImplementedError("This backbone is not implemented yet.")     # run_3dhp.py:234-235
        config.model.backbone["type"] = "hrnet_32" if width == 32 else "hrnet_48"
        pf = config.model.poseformer
        # this variant's PoseTransformer builds `config.depth` blocks per group (pose_dformer.py:199, 217-227): the engine reads it
        # from the same key (capf_config.depth; inference plans for depth != levels, the shipped configuration has 4 == 4)
        super().__init__(config, device, compute_dtype=compute_dtype, context_blocks=False)

    def forward(self, images, keypoints_2d_cpn, keypoints_2d_cpn_crop):
        x = super().forward(images, keypoints_2d_cpn, keypoints_2d_cpn_crop)       # [B, 1, 17, 3]
        b, _, p, _ = x.shape
        return x.view(b, 1, p, 3, 1).permute(0, 3, 1, 2, 4).contiguous(), None


def mpi_preset(cfg, backbone):
    """The per-backbone patch of ContextPose_mpi/run_3dhp.py:219-235 on top of common/cfg.py's defaults."""
    cfg.model.backbone.type = backbone
    cfg.model.backbone.fix_weights = True
    cfg.model.poseformer.depth = 4
    cfg.model.poseformer.levels = 4
    if backbone == "hrnet_48":
        cfg.model.backbone.STAGE2.NUM_CHANNELS = [48, 96]
        cfg.model.backbone.STAGE3.NUM_CHANNELS = [48, 96, 192]
        cfg.model.backbone.STAGE4.NUM_CHANNELS = [48, 96, 192, 384]
        cfg.model.poseformer.base_dim = 48
        cfg.model.poseformer.embed_dim_ratio = 96
    elif backbone == "hrnet_32":
        cfg.model.backbone.STAGE2.NUM_CHANNELS = [32, 64]
        cfg.model.backbone.STAGE3.NUM_CHANNELS = [32, 64, 128]
        cfg.model.backbone.STAGE4.NUM_CHANNELS = [32, 64, 128, 256]
        cfg.model.poseformer.base_dim = 32
        cfg.model.poseformer.embed_dim_ratio = 64
    else:
        raise NotImplementedError("This backbone is not implemented yet.")
    return cfg
