"""Host-side mirror of the sibling application's model package (ContextPose_mpi/model): the MPI-INF-3DHP
variant of the hot path — same HRNet backbone, PoseTransformer WITHOUT the deformable
context blocks, embed_dim_ratio 64 (W32) / 96 (W48)."""
