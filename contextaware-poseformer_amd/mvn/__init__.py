"""Host-side mirror of the reference's `mvn` package for the hot path only:
`from mvn.models.conpose import CA_PF`, `from mvn.utils.cfg import config, update_config`,
`from mvn.models.loss import MPJPE` work as they do against ContextPose/mvn, but CA_PF.forward runs
on libcapf.so (hand-written gfx950 kernels) instead of torch.nn ops."""
