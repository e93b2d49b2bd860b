"""GPU, operator level: the stateless C-ABI operators (capf_op_conv / capf_op_linear: the fp32-MFMA
implicit GEMM behind every conv and nn.Linear of the path) against plain PyTorch fp32 (CPU) on the
same inputs.  Covers every shape class of SURVEY.md Appendix A plus ragged tiles, stride 2, 1x1, 7x7,
Cin=3 stem, channel counts that are not multiples of 32, residual / ReLU / GELU epilogues."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (Cin, Cout, ks, stride, H, W, B, act, residual, bn)
CONV_CASES = [
    (32, 32, 3, 1, 64, 64, 2, 1, True, True),      # branch-0 BasicBlock conv2
    (64, 64, 3, 1, 32, 32, 3, 1, False, True),
    (128, 128, 3, 1, 16, 16, 2, 0, True, False),
    (256, 256, 3, 1, 8, 8, 5, 1, True, True),
    (3, 64, 3, 2, 64, 48, 2, 1, False, True),      # stem (small-Cin loader)
    (3, 64, 7, 2, 96, 72, 1, 1, False, True),      # CPN stem 7x7
    (64, 256, 1, 1, 16, 12, 2, 1, True, True),     # bottleneck expand
    (256, 64, 1, 1, 16, 12, 2, 1, False, True),
    (32, 64, 3, 2, 16, 12, 3, 0, False, True),     # fuse down path
    (48, 48, 3, 1, 24, 20, 1, 1, True, True),      # W48: K = 432 is not a multiple of 32
    (96, 48, 1, 1, 10, 6, 2, 0, False, True),
    (48, 96, 3, 2, 12, 10, 2, 1, False, True),
    (2048, 256, 1, 1, 4, 3, 1, 1, False, True),    # CPN lateral
    (512, 512, 3, 2, 12, 10, 1, 1, False, False),
    (256, 17, 3, 1, 9, 7, 1, 0, False, False),     # ragged N = 17
    (32, 32, 3, 1, 5, 3, 7, 1, True, False),       # tiny ragged image, M = 105
    (64, 256, 1, 1, 64, 64, 50, 1, True, True),    # layer1 conv3 + residual at full size (128x128 tiles, M = 204800)
]


@pytest.mark.parametrize("ci,co,ks,st,H,W,B,act,res,bn", CONV_CASES)
def test_conv_bn_act_matches_torch(ci, co, ks, st, H, W, B, act, res, bn):
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci * 131 + co * 7 + ks + st + H)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5
    bnp = None
    if bn:
        bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1,
               torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) * 0.4 + 0.8)
    want = F.conv2d(x, w, None, st, ks // 2)
    if bn:
        want = F.batch_norm(want, bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    r = torch.randn_like(want) if res else None
    if res:
        want = want + r
    if act == 1:
        want = F.relu(want)
    wp, bias = capf.pack_conv(w.cuda(), tuple(t.cuda() for t in bnp) if bn else None)
    got = capf.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, ks, st, act,
                         r.permute(0, 2, 3, 1).contiguous().cuda() if res else None)
    got = got.cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,act,res", [(1088, 1920, 640, 0, False), (1088, 640, 1280, 0, True), (85, 384, 128, 0, False),
                                           (4352, 256, 128, 2, False), (272, 32, 32, 0, True), (17, 16, 128, 0, False),
                                           (300, 48, 128, 0, False), (1, 640, 640, 2, True), (129, 33, 64, 0, False)])
def test_linear_matches_torch(M, N, K, act, res):
    from capf import lib as capf
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    want = F.linear(x, w, b)
    r = torch.randn(M, N, generator=g) if res else None
    if res:
        want = want + r
    if act == 2:
        want = F.gelu(want)
    got = capf.linear(x.cuda(), w.cuda(), b.cuda(), act, r.cuda() if res else None).cpu()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


WINO_CASES = [(64, 64, 32, 32, 3, 1, True), (32, 32, 64, 64, 2, 1, False), (128, 128, 16, 16, 5, 1, True), (256, 256, 8, 8, 3, 0, True),
              (64, 64, 64, 64, 1, 1, True), (96, 96, 6, 10, 2, 0, False), (64, 128, 12, 8, 3, 1, True), (128, 32, 16, 12, 2, 1, False),
              (32, 64, 2, 2, 1, 1, True), (256, 128, 24, 18, 2, 1, True)]


@pytest.mark.parametrize("ci,co,H,W,B,act,res", WINO_CASES)
def test_winograd_conv_matches_torch(ci, co, H, W, B, act, res):
    """The F(2,3)-along-W kernel (csrc/igemm_wino.hip) against an fp32 PyTorch conv + eval BatchNorm: all three block-tile
    configurations (64x64, 64x32 for 32-channel outputs, 32x64 for few tiles), ragged tile counts, W as small as 2
    (every tile touches both borders), residual / ReLU epilogues."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci * 7 + co + H * 3 + W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
           torch.rand(co, generator=g) * 0.4 + 0.8)
    want = F.batch_norm(F.conv2d(x, w, None, 1, 1), bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    r = torch.randn_like(want) if res else None
    if res:
        want = want + r
    if act == 1:
        want = F.relu(want)
    wp, bias = capf.pack_conv_wino(w.cuda(), tuple(t.cuda() for t in bnp))
    got = capf.conv_nhwc_wino(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                              r.permute(0, 2, 3, 1).contiguous().cuda() if res else None).cpu().permute(0, 3, 1, 2)
    err = (got - want).abs().max().item()
    assert err < 3e-5 * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("ci,co,H,W,B,act,res", [(64, 64, 32, 32, 2, 1, True), (32, 32, 16, 64, 3, 1, False), (128, 128, 8, 8, 5, 0, True),
                                                 (96, 96, 6, 12, 2, 1, True), (256, 64, 4, 4, 3, 1, False), (64, 160, 12, 8, 2, 1, True)])
def test_winograd_f43_conv_matches_torch(ci, co, H, W, B, act, res):
    """The F(4,3)-along-W kernel (four-pixel tiles, six positions) — the plan's default for every 3x3 stride-1 fp32 conv of
    HRNet whose row length is a multiple of 4 (csrc/plan.cpp conv_bn), on the three-resident 16-channel-superchunk tile
    (igemm_wino.hip wino43s_tile): same comparison, tolerance 1e-4 relative (its transform constants reach 8 and 1/24;
    measured 1.3e-5 .. 2.9e-5).  Production tile counts: tests/test_gpu_layerwise.py (every conv of the plan at B = 64 / 512)."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci + co * 5 + H + W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
           torch.rand(co, generator=g) * 0.4 + 0.8)
    want = F.batch_norm(F.conv2d(x, w, None, 1, 1), bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    r = torch.randn_like(want) if res else None
    if res:
        want = want + r
    if act == 1:
        want = F.relu(want)
    wp, bias = capf.pack_conv_wino(w.cuda(), tuple(t.cuda() for t in bnp), variant=43)
    got = capf.conv_nhwc_wino(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                              r.permute(0, 2, 3, 1).contiguous().cuda() if res else None).cpu().permute(0, 3, 1, 2)
    err = (got - want).abs().max().item()
    assert err < 1e-4 * max(1.0, want.abs().max().item()), err


def test_grouped_winograd_launch_is_bit_identical_to_single_launches():
    from capf import lib as capf
    g = torch.Generator().manual_seed(3)
    probs = []
    for c, r_ in ((32, 32), (64, 16), (128, 8), (256, 4)):
        x = torch.randn(6, r_, r_, c, generator=g).cuda()
        w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).cuda()
        res = torch.randn(6, r_, r_, c, generator=g).cuda()
        wp, b = capf.pack_conv_wino(w)
        probs.append((x, wp, b, 1, res))
    grouped = capf.conv_nhwc_wino_group(probs)
    for (x, wp, b, act, res), yg in zip(probs, grouped):
        assert torch.equal(yg, capf.conv_nhwc_wino(x, wp, b, act, res))


def test_grouped_winograd_f43_launch_is_bit_identical_to_single_launches():
    """F(4,3) problems of a level share the three-resident grid (igemm_wino43_group_kernel); a mixed level (one row length
    not a multiple of 4 -> that conv stays on F(2,3)) is split into the two grids.  Either way every conv sums exactly as on
    its own."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(5)
    probs = []
    for c, r_ in ((32, 32), (64, 16), (128, 8), (256, 4)):
        x = torch.randn(6, r_, r_, c, generator=g).cuda()
        w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).cuda()
        res = torch.randn(6, r_, r_, c, generator=g).cuda()
        wp, b = capf.pack_conv_wino(w, variant=43)
        probs.append((x, wp, b, 1, res))
    grouped = capf.conv_nhwc_wino_group(probs)
    for (x, wp, b, act, res), yg in zip(probs, grouped):
        assert torch.equal(yg, capf.conv_nhwc_wino(x, wp, b, act, res))


@pytest.mark.parametrize("M,N,K,gelu,res", [(1088, 1920, 640, False, False), (1088, 640, 1280, False, True), (5440, 384, 128, False, False),
                                            (4352, 256, 128, True, False), (17, 640, 640, False, True), (85, 128, 256, False, True),
                                            (4352, 1280, 640, True, False), (300, 132, 64, False, False)])
def test_linear_bf16_matches_torch_on_bf16_rounded_operands(M, N, K, gelu, res):
    """The lifter's projections on the bf16 MFMA path (compute_dtype = bf16): bf16 operands, fp32 accumulation, fp32 result
    (+ fp32 residual) or GELU + one bf16 rounding — against fp32 F.linear of the SAME bf16-rounded operands."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, generator=g)
    want = F.linear(x.float(), w.float(), b)
    r = torch.randn(M, N, generator=g) if res else None
    if res:
        want = want + r
    got = capf.linear_bf16(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, gelu=gelu)
    if gelu:
        want = F.gelu(want)
        assert got.dtype == torch.bfloat16
        tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
    else:
        assert got.dtype == torch.float32
        tol = 3e-5 * max(1.0, want.abs().max().item())
    assert (got.float().cpu() - want).abs().max().item() <= tol


BF16_CASES = [(32, 32, 3, 1, 64, 64, 2, 1, True), (64, 64, 3, 1, 32, 32, 2, 1, False), (48, 48, 3, 1, 24, 20, 1, 1, True),
              (256, 256, 3, 1, 8, 8, 3, 0, True), (64, 256, 1, 1, 16, 12, 2, 1, True), (96, 48, 1, 1, 10, 6, 2, 0, False),
              (32, 64, 3, 2, 16, 12, 3, 1, False), (2048, 256, 1, 1, 4, 3, 1, 1, False), (128, 128, 3, 1, 5, 3, 7, 1, True),
              # Cout % 8 != 0: the epilogue's element-wise tail instead of its 16-byte path
              (32, 36, 3, 1, 9, 7, 2, 1, True), (64, 20, 1, 1, 6, 5, 3, 0, False),
              # >= 2048 tiles per launch: the ping-pong (one-stage) schedule, 128x64 and 128x128 tiles
              (48, 48, 3, 1, 64, 64, 66, 1, True), (64, 128, 1, 1, 64, 64, 40, 1, True), (32, 32, 3, 1, 64, 64, 65, 0, False)]


@pytest.mark.parametrize("ci,co,ks,st,H,W,B,act,res", BF16_CASES)
def test_conv_bf16_matches_torch_on_bf16_rounded_operands(ci, co, ks, st, H, W, B, act, res):
    """bf16 MFMA conv == fp32 conv of the SAME bf16-rounded inputs/weights (products of two bf16 numbers are
    exact in fp32, so only accumulation order and the final bf16 rounding differ)."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci + co * 3 + ks + H)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16()
    w = (torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5)
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
           torch.rand(co, generator=g) * 0.4 + 0.8)
    wp, bias = capf.pack_conv_bf16(w.cuda(), tuple(t.cuda() for t in bnp))
    kreal = ks * ks * ci
    w_fold = wp[:, :kreal].float().cpu().view(co, ks, ks, ci).permute(0, 3, 1, 2).contiguous()   # what the kernel multiplies
    want = F.conv2d(x.float(), w_fold, bias.cpu(), st, ks // 2)
    r = torch.randn_like(want).bfloat16() if res else None
    if res:
        want = want + r.float()
    if act == 1:
        want = F.relu(want)
    got = capf.conv_nhwc_bf16(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, ks, st, act,
                              r.permute(0, 2, 3, 1).contiguous().cuda() if res else None)
    got = got.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5          # one bf16 rounding of the result
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("backbone,mode", [("hrnet_32", 0), ("hrnet_32", 1), ("hrnet_32", 2), ("cpn", 2), ("cpn", 1)])
def test_preprocess_kernel_is_bit_exact(backbone, mode):
    """N1: capf_preprocess == the prefetcher's torch expressions (restated in the oracle), bit for bit."""
    import capf_oracle as oracle
    from capf import lib as capf
    g = torch.Generator().manual_seed(mode + len(backbone))
    B, H, W = 3, 256, 192
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    gt = torch.randn(B, 1, 17, 3, generator=g)
    k2d = torch.rand(B, 17, 2, generator=g) * 2 - 1
    kc = torch.rand(B, 17, 2, generator=g) * 191
    want = oracle.prefetch_preprocess(img, gt, k2d, kc, backbone, is_train=(mode == 1), flip=(mode == 1), flip_test=(mode == 2))
    got = capf.preprocess(img.cuda(), gt.cuda(), k2d.cuda(), kc.cuda(), backbone, mode)
    wi, wg, wk, wc = want
    if mode == 2:      # the reference stacks on dim 1 ([B,2,...]); the kernel emits [2,B,...] so ONE forward serves both
        wi, wk, wc = wi.transpose(0, 1), wk.transpose(0, 1), wc.transpose(0, 1)
    assert torch.equal(got[0].cpu(), wi.contiguous())
    assert torch.equal(got[1].cpu(), wg)
    assert torch.equal(got[2].cpu(), wk.contiguous())
    assert torch.equal(got[3].cpu(), wc.contiguous())


def test_fliptest_fusion_is_bit_exact_and_end_to_end():
    """N2: the fused un-mirror + average == train.py:177-180; and one 2B forward == two B forwards."""
    import capf_oracle as oracle
    from capf import lib as capf
    p = torch.randn(2, 5, 1, 17, 3)
    want = oracle.fliptest_fuse(p[0], p[1])
    assert torch.equal(capf.fliptest_fuse(p.cuda()).cpu(), want)


def test_flip_test_single_forward_equals_two_forwards_and_prefetcher_runs():
    """N1+N2 end to end: prefetcher mirror (flip-test mode) -> one 2B forward + fused un-mirror/average ==
    the reference's two forwards + torch fusion (computed here with the oracle's restatement of the fusion)."""
    import capf_oracle as oracle
    from conftest import make_model
    from mvn.datasets.utils import data_prefetcher
    g = torch.Generator().manual_seed(5)
    B = 2
    batch = (torch.randint(0, 256, (B, 256, 192, 3), generator=g, dtype=torch.uint8), torch.randn(B, 1, 17, 3, generator=g),
             torch.rand(B, 17, 2, generator=g) * 2 - 1, torch.rand(B, 17, 2, generator=g) * 191)
    pf = data_prefetcher([batch], torch.device("cuda"), is_train=False, flip_test=True, backbone="hrnet_32")
    images, gt, k2d, kc = pf.next()
    assert pf.next() is None and images.shape == (B, 2, 256, 192, 3)
    model, _ = make_model("hrnet_32", device="cuda", wseed=9)
    with torch.no_grad():
        a = model(images[:, 0], k2d[:, 0], kc[:, 0].clone())              # train.py:171-176
        b = model(images[:, 1], k2d[:, 1], kc[:, 1].clone())
        want = oracle.fliptest_fuse(a.cpu(), b.cpu())
        got = model.forward_flip_test(images.transpose(0, 1).contiguous(), k2d.transpose(0, 1).contiguous(),
                                      kc.transpose(0, 1).contiguous().clone())
    assert (got.cpu() - want).abs().max().item() < 1e-5


# ---- grouped launch (capf_op_conv_group): independent convs of one dependency level in ONE grid ----------
GROUPS = [
    # the four branches of an HRNet-W32 stage-4 module at the same depth (conv2 of a BasicBlock: residual + ReLU)
    [(32, 32, 3, 1, 64, 64, 2, 1, True), (64, 64, 3, 1, 32, 32, 2, 1, True), (128, 128, 3, 1, 16, 16, 2, 1, True),
     (256, 256, 3, 1, 8, 8, 2, 1, True)],
    # a fuse layer: 1x1 convs at low resolution + the first stride-2 convs of the down paths, ragged sizes
    [(64, 32, 1, 1, 12, 10, 3, 0, False), (128, 32, 1, 1, 6, 5, 3, 0, False), (32, 32, 3, 2, 24, 20, 3, 1, False),
     (32, 64, 3, 2, 24, 20, 3, 0, False), (48, 96, 3, 2, 12, 10, 1, 0, False), (256, 17, 3, 1, 9, 7, 1, 0, False),
     (32, 32, 3, 1, 5, 3, 7, 1, True), (96, 48, 1, 1, 10, 6, 2, 0, False)],
    # two problems, one of them a single tile
    [(64, 64, 3, 1, 8, 8, 1, 1, False), (32, 32, 3, 1, 64, 48, 5, 1, True)],
]


@pytest.mark.parametrize("gi", range(len(GROUPS)))
def test_grouped_conv_launch_is_bit_identical_to_single_launches(gi):
    from capf import lib as capf
    probs, singles = [], []
    for k, (ci, co, ks, st, H, W, B, act, res) in enumerate(GROUPS[gi]):
        g = torch.Generator().manual_seed(1000 * gi + k)
        x = torch.randn(B, H, W, ci, generator=g).cuda()
        w = (torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5).cuda()
        wp, bias = capf.pack_conv(w)
        pad = ks // 2
        ho, wo = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
        r = torch.randn(B, ho, wo, co, generator=g).cuda() if res else None
        probs.append((x, wp, bias, ks, st, act, r))
        singles.append(capf.conv_nhwc(x, wp, bias, ks, st, act=act, residual=r))
    outs = capf.conv_nhwc_group(probs)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(outs, singles)):
        assert torch.equal(a, b), f"problem {k} of group {gi} differs from its single launch"


@pytest.mark.parametrize("dtype,backbone", [("fp32", "hrnet_32"), ("bf16", "hrnet_32"), ("fp32", "cpn"), ("bf16", "hrnet_48")])
def test_engine_modes_are_bit_identical(dtype, backbone):
    """capf_set_lanes 0 / 2 (program order, grouped launches) give the same bits; mode 1 (side streams, kept for A/B runs)
    has no split-K scratch — concurrent launches would share it — so its long-K small-batch convs sum K in one pass instead
    of slices: equal to roundoff, not to the bit."""
    import copy, contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=dtype).eval()
    synth.load_synthetic(model, seed=3, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(3, 128, 96, seed=5, crop_range=(96, 128))
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    outs = []
    with torch.no_grad():
        for mode in (0, 1, 2):
            model.engine_for(img).set_lanes(mode)
            outs.append(model(img, k2d, kc.clone()).clone())
    assert torch.equal(outs[0], outs[2])
    assert (outs[0] - outs[1]).abs().max().item() <= (2e-6 if dtype == "fp32" else 1e-2)


@pytest.mark.parametrize("dtype,backbone,B,H,W,flags", [("fp32", "hrnet_32", 24, 256, 192, 0), ("bf16", "hrnet_48", 24, 256, 192, 0),
                                                         ("bf16", "cpn", 24, 256, 192, 0), ("fp32", "hrnet_32", 16, 64, 64, 2),
                                                         ("fp32", "hrnet_32", 48, 256, 192, 0)])
def test_two_chain_schedule_is_bit_identical_to_one_chain(dtype, backbone, B, H, W, flags):
    """capf_set_lanes 3 (the default at batch 16..256: a region's lanes as two grouped chains on two streams, fork / join with
    events) against mode 2 (one chain on the caller's stream) at batch 24: same kernels on the same operands, only their
    grouping and their stream differ -> the same bits, call after call (a missing event dependency would show up as a
    run-to-run difference).  The 64 x 64 / batch 16 case runs with CAPF_PLAN_NO_WINOGRAD: its 16 x 16 ... 2 x 2 maps are a handful of
    tiles per conv, so the 3x3 convs of BOTH concurrent chains take the split-K path (a conv splits by its shape alone) and must not
    share slabs and counters (each chain has its own).  Batch 48 fp32: the branch convs run on the
    split-fp32 tile (igemm_f32x3_ws.hip) in both schedules."""
    import copy, contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype=dtype, plan_flags=flags).eval()
    synth.load_synthetic(model, seed=3, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=5, crop_range=(W, H))
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    eng_out = {}
    with torch.no_grad():
        for mode in (2, 3, 3, 2, 3):
            eng = model.engine_for(img)
            eng.set_lanes(mode)
            out = model(img, k2d, kc.clone()).clone()
            maps = [eng.tensor(f"feat{l}").clone() for l in range(4)]
            if not eng_out:
                eng_out = {"out": out, "maps": maps}
            assert torch.equal(out, eng_out["out"]), f"mode {mode}"
            for a, b in zip(maps, eng_out["maps"]):
                assert torch.equal(a, b), f"mode {mode}"


def test_pointwise_chain_is_bit_identical_to_its_two_launches():
    """igemm_f32_pwchain (layer1's conv3 -> next conv1 as one launch, the first conv's accumulators feeding the second conv's
    MFMAs from registers) against the same plan with CAPF_PLAN_NO_PWCHAIN: same K order, same ((acc + bias) + res) epilogue ->
    the same bits in the four context maps and in the joints, at batch 64 (8 tiles per wave), at 41 frames and on 256 x 192 crops."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_NO_PWCHAIN
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    models = []
    for flags in (0, PLAN_NO_PWCHAIN):
        with contextlib.redirect_stdout(io.StringIO()):
            m = CA_PF(cfg, compute_dtype="fp32", plan_flags=flags).eval()
        synth.load_synthetic(m, seed=3, bn_mode="random")
        models.append(m.cuda())
    for B, W in ((64, 256), (41, 256), (48, 192)):                # (256 x 192: the reference's own crop, 64 x 48 maps)
        img, k2d, kc = synth.synth_inputs(B, 256, W, seed=5 + B, crop_range=(W, 256))
        img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
        res = []
        with torch.no_grad():
            for m in models:
                out = m(img, k2d, kc.clone()).clone()
                eng = m.engine_for(img)
                res.append((out, [eng.tensor(f"feat{l}")[:B].clone() for l in range(4)],
                            {n: k for n, k, _ in eng.op_table(B)}["backbone.layer1.1.conv3"]))
        assert res[0][2].startswith("igemm_f32_pwchain") and res[1][2].startswith("igemm_f32_pw<")
        assert torch.equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("backbone,B,H,W", [("hrnet_48", 64, 256, 256), ("cpn", 32, 384, 288)])
def test_bf16_pointwise_chain_agrees_with_its_two_launches(backbone, B, H, W):
    """The bf16 twin (igemm_bf16_pwchain: y rounded once in registers, the second conv's 16-product groups summed in a permuted slot
    order) against CAPF_PLAN_NO_PWCHAIN.  The first conv is bit-identical; the second differs by fp32 summation order inside an MFMA,
    i.e. by at most one bf16 rounding of its output -- which a deep bf16 network amplifies to its rounding-noise floor (bf16_report.py),
    so the context maps are held to that floor (the layer-wise test checks the chained ops themselves one by one).
    Since round 6 every layer1 bottleneck of a bf16 plan is one fused launch (bneck_bf16.hip), so the chained pairs exist only under
    CAPF_PLAN_NO_BNECK: both plans of this test carry that flag."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_NO_BNECK, PLAN_NO_PWCHAIN
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), backbone)
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=21, crop_range=(W, H))
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    res = []
    for flags in (PLAN_NO_BNECK, PLAN_NO_BNECK | PLAN_NO_PWCHAIN):
        with contextlib.redirect_stdout(io.StringIO()):
            m = CA_PF(cfg, compute_dtype="bf16", plan_flags=flags).eval()
        synth.load_synthetic(m, seed=3, bn_mode="random")
        m = m.cuda()
        with torch.no_grad():
            m(img, k2d, kc.clone())
        eng = m.engine_for(img)
        res.append(([eng.tensor(f"feat{l}")[:B].float().clone() for l in range(4)], [k for _, k, _ in eng.op_table(B)]))
    assert any(k.startswith("igemm_bf16_pwchain") for k in res[0][1]) and not any(k.startswith("igemm_bf16_pwchain") for k in res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"bf16 chain vs two launches ({backbone}): relative L2 {rel:.2e}")
        assert rel < 1.5e-2


def test_grouped_bf16_conv_launch_is_bit_identical_to_single_launches():
    """The engine's bf16 grouped launch (igemm_bf16_group_kernel) against single bf16 launches: run the
    backbone of a bf16 model with capf_set_lanes 0 and 2 and compare the four context maps bit for bit."""
    import copy, contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, compute_dtype="bf16").eval()
    synth.load_synthetic(model, seed=4, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(2, 256, 192, seed=6)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps = []
    with torch.no_grad():
        for mode in (0, 2):
            eng = model.engine_for(img)
            eng.set_lanes(mode)
            model(img, k2d, kc.clone())
            maps.append([eng.tensor(f"feat{l}").clone() for l in range(4)])
    for a, b in zip(*maps):
        assert torch.equal(a, b)


def test_row_halo_engine_path_agrees_with_the_direct_bf16_kernels():
    """Batch 48 HRNet-32 bf16: the grouped branch launches have >= 2048 tiles, so the engine runs the row-halo kernel
    (capf_forward_profile_variants reports igemm_bf16_group_rh_kernel for them); a second engine planned with
    plan_flags = CAPF_PLAN_NO_ROW_HALO runs the direct kernels.  Different K order and bias placement, same bf16 operands: the four context
    maps agree to the noise two bf16 evaluations with different summation orders accumulate over ~50 layers (measured 5-8e-3,
    the same size as either run's distance from the fp32 oracle)."""
    import copy, contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(48, 256, 256, seed=16)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps, variants = [], []
    from capf.lib import PLAN_NO_ROW_HALO, PLAN_NO_WS
    for flags in (PLAN_NO_WS, PLAN_NO_WS | PLAN_NO_ROW_HALO):
        with contextlib.redirect_stdout(io.StringIO()):
            model = CA_PF(cfg, compute_dtype="bf16", plan_flags=flags).eval()
        synth.load_synthetic(model, seed=4, bn_mode="random")
        model = model.cuda()
        with torch.no_grad():
            out = model(img, k2d, kc.clone())
            eng = model.engine_for(img)
            eng.forward_profile_launches(img, k2d, kc.clone(), torch.empty_like(out), torch.cuda.current_stream().cuda_stream)
        variants.append(set(v for v in eng.profile_variants() if v >= 0))
        maps.append([eng.tensor(f"feat{l}").float().clone() for l in range(4)])
    assert 2 in variants[0] and 2 not in variants[1]
    for a, b in zip(*maps):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"row-halo vs direct: relative L2 {rel:.2e}")
        assert rel < 1.5e-2


def test_conv_fuzz_against_torch():
    """Seeded random conv shapes (ragged M and N, odd image sizes, stride 1 / 2, 1x1 / 3x3 / 5x5, channel counts
    that are and are not multiples of 32, with and without bias / residual / ReLU) against PyTorch on the CPU, and
    the same problems again as grouped launches (bit-identical to the single launches)."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20260928)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    probs, singles = [], []
    for case in range(36):
        ci = 4 * ri(1, 40)
        co = 4 * ri(1, 40) if case % 5 else ri(1, 67)          # every fifth case: N not a multiple of 4
        ks = (1, 3, 3, 5)[ri(0, 3)]
        st = ri(1, 2)
        H, W, B = ri(1, 23), ri(1, 19), ri(1, 6)
        act, res = ri(0, 1), bool(ri(0, 1))
        x = torch.randn(B, ci, H, W, generator=rng)
        w = torch.randn(co, ci, ks, ks, generator=rng) / (ci * ks * ks) ** 0.5
        bn = (torch.rand(co, generator=rng) + 0.5, torch.randn(co, generator=rng) * 0.1,
              torch.randn(co, generator=rng) * 0.1, torch.rand(co, generator=rng) * 0.4 + 0.8)
        want = F.batch_norm(F.conv2d(x, w, None, st, ks // 2), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
        r = torch.randn_like(want) if res else None
        if res:
            want = want + r
        if act:
            want = F.relu(want)
        wp, bias = capf.pack_conv(w.cuda(), tuple(t.cuda() for t in bn))
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        rd = r.permute(0, 2, 3, 1).contiguous().cuda() if res else None
        got = capf.conv_nhwc(xd, wp, bias, ks, st, act, rd)
        err = (got.cpu().permute(0, 3, 1, 2) - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (case, ci, co, ks, st, H, W, B, err)
        if co % 4 == 0:
            probs.append((xd, wp, bias, ks, st, act, rd))
            singles.append(got)
    for i in range(0, len(probs), 8):
        outs = capf.conv_nhwc_group(probs[i:i + 8])
        for a, b in zip(outs, singles[i:i + 8]):
            assert torch.equal(a, b)


def test_linear_fuzz_against_torch():
    """Seeded random (M, N, K) with K % 32 == 0, ragged M / N, bias / residual / ReLU / GELU against PyTorch (CPU)."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(777)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    for case in range(24):
        M, N, K = ri(1, 700), ri(1, 300), 32 * ri(1, 20)
        act, res, has_b = ri(0, 2), bool(ri(0, 1)), bool(ri(0, 1))
        x = torch.randn(M, K, generator=rng)
        w = torch.randn(N, K, generator=rng) / K ** 0.5
        b = torch.randn(N, generator=rng) if has_b else None
        r = torch.randn(M, N, generator=rng) if res else None
        want = F.linear(x, w, b)
        if res:
            want = want + r                      # epilogue order of the kernel: + bias, + residual, activation
        if act == 2:
            want = F.gelu(want)
        if act == 1:
            want = F.relu(want)
        got = capf.linear(x.cuda(), w.cuda(), b.cuda() if has_b else None, act=act, residual=r.cuda() if res else None).cpu()
        err = (got - want).abs().max().item()
        assert err < 3e-5 * max(1.0, want.abs().max().item()), (case, M, N, K, act, res, has_b, err)


def test_abi_error_codes_on_a_device_handle():
    """Every failure is a negative status + capf_last_error text, never an exception across the ABI or a crash:
    forward before the parameters were packed, batch outside 1..max_batch, workspace too small, unknown
    parameter name, parameter shape mismatch."""
    import copy, ctypes
    from capf import CapfError, Engine
    from mvn.models import _native
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    c = _native.make_capf_config(cfg, 128, 96)
    c.max_batch = 4
    eng = Engine(c, device=0)
    lib, h = eng.lib, eng.h
    img = torch.zeros(2, 128, 96, 3, device="cuda")
    k2d = torch.zeros(2, 17, 2, device="cuda")
    kc = torch.zeros(2, 17, 2, device="cuda")
    out = torch.zeros(2, 1, 17, 3, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fwd = lambda b: lib.capf_forward(h, None, P(img), P(k2d), P(kc), b, P(out))
    assert fwd(2) < 0 and b"capf_params_changed" in lib.capf_last_error(h)         # nothing bound / packed yet
    shp = (ctypes.c_int64 * 4)(3, 3, 3, 3)
    assert lib.capf_set_param(h, b"backbone.no_such.weight", P(img), shp, 4) < 0
    assert b"unknown parameter" in lib.capf_last_error(h)
    assert lib.capf_set_param(h, b"backbone.conv1.weight", P(img), shp, 4) < 0      # [64, 3, 3, 3] expected
    assert b"shape mismatch" in lib.capf_last_error(h)
    # a fully bound model: batch range and workspace checks
    import contextlib, io
    from capf import synth
    from mvn.models.conpose import CA_PF
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg).eval()
    synth.load_synthetic(model, seed=5, bn_mode="random")
    model = model.cuda()
    with torch.no_grad():
        model(img, k2d, kc.clone())
    e2 = model.engine_for(img)
    f2 = lambda b: e2.lib.capf_forward(e2.h, None, P(img), P(k2d), P(kc), b, P(out))
    assert f2(0) < 0 and f2(10 ** 6) < 0 and b"batch out of range" in e2.lib.capf_last_error(e2.h)
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    assert e2.lib.capf_set_workspace(e2.h, P(small), small.numel()) == 0
    assert f2(2) < 0 and b"workspace" in e2.lib.capf_last_error(e2.h)
    with pytest.raises(CapfError):
        e2._check(f2(2), "forward")


RH_CASES = [(64, 64, 32, 32, 2, 1, True), (48, 48, 24, 20, 1, 1, True), (32, 32, 16, 12, 3, 0, False), (96, 96, 9, 7, 2, 1, True),
            (128, 256, 5, 3, 7, 1, False), (192, 192, 8, 8, 2, 1, True), (64, 36, 7, 5, 2, 1, True), (48, 48, 64, 64, 3, 1, True),
            (32, 32, 1, 1, 5, 1, True), (64, 64, 2, 130, 1, 0, False)]


@pytest.mark.parametrize("ci,co,H,W,B,act,res", RH_CASES)
def test_conv_bf16_row_halo_matches_torch_on_bf16_rounded_operands(ci, co, H, W, B, act, res):
    """The row-halo 3x3 kernel (one staged activation tile for the three kw taps, border taps zeroed in registers) against
    fp32 F.conv2d of the SAME bf16-rounded operands; tiles of 126 pixels straddle image rows and images in every case."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci + co * 3 + H * 7 + W)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16()
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
           torch.rand(co, generator=g) * 0.4 + 0.8)
    wp, bias, cw = capf.pack_conv_bf16_rh(w.cuda(), tuple(t.cuda() for t in bnp))
    assert cw == (64 if ci % 64 == 0 else 48 if ci % 48 == 0 else 32)
    w_fold = wp.float().cpu().view(co, 3, ci // cw, 3, cw).permute(0, 2, 4, 1, 3).reshape(co, ci, 3, 3)   # what the kernel multiplies
    want = F.conv2d(x.float(), w_fold, bias.cpu(), 1, 1)
    r = torch.randn_like(want).bfloat16() if res else None
    if res:
        want = want + r.float()
    if act == 1:
        want = F.relu(want)
    got = capf.conv_nhwc_bf16_rh(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                                 r.permute(0, 2, 3, 1).contiguous().cuda() if res else None)
    got = got.float().cpu().permute(0, 3, 1, 2)
    tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("ci,co,res", [(64, 256, True), (256, 64, False), (96, 72, True), (32, 8, False)])
def test_pointwise_fp32_kernel_is_bit_identical_to_the_general_kernel(ci, co, res):
    """igemm_f32_pw (1x1 convs with >= 2048 tiles: ping-pong schedule, coalesced epilogue) against the general implicit GEMM:
    the same frames in a batch small enough to stay on the general kernel give the same bits, and both match F.conv2d."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(ci * 7 + co)
    H = W = 64
    B = 66 if co <= 64 else 20                     # ceil(B*4096/128) * ceil(co/64) >= 2048
    x = torch.randn(B, H, W, ci, generator=g)
    w = torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
           torch.rand(co, generator=g) * 0.4 + 0.8)
    r = torch.randn(B, H, W, co, generator=g) if res else None
    wp, bias = capf.pack_conv(w.cuda(), tuple(t.cuda() for t in bnp))
    big = capf.conv_nhwc(x.cuda(), wp, bias, 1, 1, 1, r.cuda() if res else None)
    nb = 3                                          # 3 frames: 96 row tiles, the general kernel
    small = capf.conv_nhwc(x[:nb].contiguous().cuda(), wp, bias, 1, 1, 1, r[:nb].contiguous().cuda() if res else None)
    assert torch.equal(big[:nb], small)
    w_fold = wp[:, :ci].cpu().view(co, ci, 1, 1)
    want = F.conv2d(x.permute(0, 3, 1, 2), w_fold, bias.cpu())
    if res:
        want = want + r.permute(0, 3, 1, 2)
    want = F.relu(want).permute(0, 2, 3, 1)
    assert (big.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("chans,B", [((48, 96, 192, 384), 40), ((32, 64, 128, 256), 40), ((48, 96), 6)])
def test_grouped_bf16_launch_with_row_halo_tiles_matches_torch(chans, B):
    """One grouped bf16 launch of the four HRNet branch convs (3x3, stride 1, residual, ReLU) the way the engine issues them:
    at B = 40 the launch has >= 2048 tiles and runs igemm_bf16_group_rh_kernel (row-halo tiles of every chunk width and column
    count, padded tile ids, mixed problems in one grid); at B = 6 it stays on the ring kernel.  Every output against fp32
    F.conv2d of the same bf16-rounded operands."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(sum(chans) + B)
    probs, wants = [], []
    for i, c in enumerate(chans):
        r = 64 >> i
        x = torch.randn(B, c, r, r, generator=g).bfloat16()
        w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
        bnp = (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
               torch.rand(c, generator=g) * 0.4 + 0.8)
        res = torch.randn(B, c, r, r, generator=g).bfloat16()
        bn_dev = tuple(t.cuda() for t in bnp)
        wp, bias = capf.pack_conv_bf16(w.cuda(), bn_dev)
        wrh, bias2, cw = capf.pack_conv_bf16_rh(w.cuda(), bn_dev)
        assert torch.equal(bias, bias2)
        w_fold = wp[:, :9 * c].float().cpu().view(c, 3, 3, c).permute(0, 3, 1, 2).contiguous()
        wants.append(F.relu(F.conv2d(x.float(), w_fold, bias.cpu(), 1, 1) + res.float()))
        probs.append((x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, 3, 1, 1, res.permute(0, 2, 3, 1).contiguous().cuda(), wrh))
    outs, variant = capf.conv_nhwc_bf16_group(probs)
    assert variant == (2 if B >= 40 else 0)
    for y, want in zip(outs, wants):
        got = y.float().cpu().permute(0, 3, 1, 2)
        tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
        assert (got - want).abs().max().item() <= tol


def test_row_halo_fuzz_against_torch():
    """Seeded random 3x3 problems through the row-halo kernel: every chunk width (Cin multiples of 32 / 48 / 64), ragged Cout
    (multiples of 4 and of 8), image sizes from 1x1 up, widths that are not multiples of anything, with and without residual /
    ReLU -- tiles of 126 flat pixels cross image rows and images in almost every case."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20260929)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    for case in range(24):
        ci = (32, 48, 64, 96, 128, 160, 192)[ri(0, 6)]
        co = 4 * ri(1, 50) if case % 3 else 8 * ri(1, 30)
        H, W, B = ri(1, 37), ri(1, 41), ri(1, 5)
        act, res = ri(0, 1), bool(ri(0, 1))
        x = torch.randn(B, ci, H, W, generator=rng).bfloat16()
        w = torch.randn(co, ci, 3, 3, generator=rng) / (ci * 9) ** 0.5
        bnp = (torch.rand(co, generator=rng) + 0.5, torch.randn(co, generator=rng) * 0.1, torch.randn(co, generator=rng) * 0.1,
               torch.rand(co, generator=rng) * 0.4 + 0.8)
        wp, bias, cw = capf.pack_conv_bf16_rh(w.cuda(), tuple(t.cuda() for t in bnp))
        w_fold = wp.float().cpu().view(co, 3, ci // cw, 3, cw).permute(0, 2, 4, 1, 3).reshape(co, ci, 3, 3)
        want = F.conv2d(x.float(), w_fold, bias.cpu(), 1, 1)
        r = torch.randn_like(want).bfloat16() if res else None
        if res:
            want = want + r.float()
        if act:
            want = F.relu(want)
        got = capf.conv_nhwc_bf16_rh(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                                     r.permute(0, 2, 3, 1).contiguous().cuda() if res else None)
        got = got.float().cpu().permute(0, 3, 1, 2)
        tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
        err = (got - want).abs().max().item()
        assert err <= tol, f"case {case}: Cin {ci} Cout {co} {H}x{W} B{B} act {act} res {res}: {err:.3e} > {tol:.3e}"


def _ws_fold(wp, co, ci):
    """Undo capf_op_pack_conv_bf16_ws: packed [slice][Cin / 16][tap][NS][quad position][8] -> folded weights [co, ci, 3, 3] (fp32)."""
    ns = 96 if co % 96 == 0 else (32 if co <= 32 else 64)
    nsl = (co + ns - 1) // ns
    t = wp.float().cpu().view(nsl, ci // 16, 9, ns, 2, 8)
    n = torch.arange(ns)
    swap = ((n >> 3) & 1).bool()
    t = torch.where(swap[None, None, None, :, None, None], t.flip(4), t)           # quad position -> channel half
    w = t.permute(0, 3, 1, 4, 5, 2).reshape(nsl * ns, ci, 9)[:co]                  # [n, (cc, half, e), tap]
    return w.reshape(co, ci, 3, 3).contiguous()


def test_ws_fuzz_against_torch():
    """Seeded random 3x3 / stride-1 problems through the 2-D halo tile (csrc/igemm_bf16_ws.hip): every channel-slice width (Cout
    <= 32, multiples of 96, everything else on 64 with padded rows), Cin multiples of 16, image sizes whose tiles are part rows,
    whole images and several images, ragged last tiles, with and without residual / ReLU -- against fp32 F.conv2d of the same
    bf16-rounded operands (the folded weights are read back from the packed layout, which also checks the pack)."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20260930)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    for case in range(28):
        ci = 16 * ri(1, 14)
        co = (8 * ri(1, 4), 96 * ri(1, 3), 8 * ri(5, 40))[case % 3]
        H, W, B = ri(1, 40), ri(1, 70), ri(1, 5)
        if case == 5:
            H, W, B = 8, 8, 9                     # several images per tile, ragged last tile
        if case == 6:
            H, W, B = 64, 64, 2                   # part rows of an image per tile
        act, res = ri(0, 1), bool(ri(0, 1))
        x = torch.randn(B, ci, H, W, generator=rng).bfloat16()
        w = torch.randn(co, ci, 3, 3, generator=rng) / (ci * 9) ** 0.5
        bnp = (torch.rand(co, generator=rng) + 0.5, torch.randn(co, generator=rng) * 0.1, torch.randn(co, generator=rng) * 0.1,
               torch.rand(co, generator=rng) * 0.4 + 0.8)
        wp, bias = capf.pack_conv_bf16_ws(w.cuda(), tuple(t.cuda() for t in bnp))
        w_fold = _ws_fold(wp, co, ci)
        want = F.conv2d(x.float(), w_fold, bias.cpu(), 1, 1)
        r = torch.randn_like(want).bfloat16() if res else None
        if res:
            want = want + r.float()
        if act:
            want = F.relu(want)
        got, = capf.conv_nhwc_bf16_ws_group([(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                                              r.permute(0, 2, 3, 1).contiguous().cuda() if res else None, co)])
        got = got.float().cpu().permute(0, 3, 1, 2)
        tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
        err = (got - want).abs().max().item()
        assert err <= tol, f"case {case}: Cin {ci} Cout {co} {H}x{W} B{B} act {act} res {res}: {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("chans,B", [((48, 96, 192, 384), 24), ((32, 64, 128, 256), 9), ((64, 128, 256, 512), 3)])
def test_grouped_ws_launch_matches_torch_and_single_launches(chans, B):
    """The four HRNet branch convs (3x3, stride 1, residual, ReLU) as ONE grouped launch of the 2-D halo tile -- slices of 32 / 64 /
    96 channels, padded tile ids, problems of different K in one grid -- against fp32 F.conv2d of the same bf16-rounded operands,
    and bit-identical to the four single launches."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(sum(chans) + B)
    probs, wants = [], []
    for i, c in enumerate(chans):
        r = 64 >> i
        x = torch.randn(B, c, r, r, generator=g).bfloat16()
        w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
        bnp = (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
               torch.rand(c, generator=g) * 0.4 + 0.8)
        res = torch.randn(B, c, r, r, generator=g).bfloat16()
        wp, bias = capf.pack_conv_bf16_ws(w.cuda(), tuple(t.cuda() for t in bnp))
        wants.append(F.relu(F.conv2d(x.float(), _ws_fold(wp, c, c), bias.cpu(), 1, 1) + res.float()))
        probs.append((x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, 1, res.permute(0, 2, 3, 1).contiguous().cuda(), c))
    outs = capf.conv_nhwc_bf16_ws_group(probs)
    for y, want, pr in zip(outs, wants, probs):
        got = y.float().cpu().permute(0, 3, 1, 2)
        tol = 2.0 ** -8 * max(1.0, want.abs().max().item()) * 1.5
        assert (got - want).abs().max().item() <= tol
        single, = capf.conv_nhwc_bf16_ws_group([pr])
        assert torch.equal(single, y)


def test_ws_engine_path_agrees_with_the_row_halo_kernels():
    """Batch 48 HRNet-32 bf16: the product plan runs the branch levels on igemm_bf16_group_ws_kernel (capf_forward_profile_variants
    reports 3), a plan with CAPF_PLAN_NO_WS on the row-halo kernel (2).  Same bf16 operands, different summation order: the four
    context maps agree to the noise two bf16 evaluations accumulate over ~50 layers."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_NO_WS
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(48, 256, 256, seed=16)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps, variants = [], []
    for flags in (0, PLAN_NO_WS):
        with contextlib.redirect_stdout(io.StringIO()):
            model = CA_PF(cfg, compute_dtype="bf16", plan_flags=flags).eval()
        synth.load_synthetic(model, seed=4, bn_mode="random")
        model = model.cuda()
        with torch.no_grad():
            out = model(img, k2d, kc.clone())
            eng = model.engine_for(img)
            eng.forward_profile_launches(img, k2d, kc.clone(), torch.empty_like(out), torch.cuda.current_stream().cuda_stream)
        variants.append(set(v for v in eng.profile_variants() if v >= 0))
        maps.append([eng.tensor(f"feat{l}").float().clone() for l in range(4)])
    assert 3 in variants[0] and 3 not in variants[1] and 2 in variants[1]
    for a, b in zip(*maps):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"2-D halo vs row-halo: relative L2 {rel:.2e}")
        assert rel < 1.5e-2



def _x3_fold(wp, co, ci):
    """Undo capf_op_pack_conv_f32x3: packed [slice][Cin / 16][piece][tap][32][quad position][8] bf16 -> folded fp32 weights [co, ci, 3, 3]
    (the three pieces of a weight add up to it exactly)."""
    ns = 32
    nsl = (co + ns - 1) // ns
    t = wp.double().cpu().view(nsl, ci // 16, 3, 9, ns, 2, 8).sum(dim=2)
    n = torch.arange(ns)
    swap = ((n >> 3) & 1).bool()
    t = torch.where(swap[None, None, None, :, None, None], t.flip(4), t)           # quad position -> channel half
    w = t.permute(0, 3, 1, 4, 5, 2).reshape(nsl * ns, ci, 9)[:co]                  # [n, (cc, half, e), tap]
    return w.reshape(co, ci, 3, 3).contiguous()


def _x3_check(got_nhwc, x, w_fold, bias, res, act, what, direct=None):
    """fp32 result of the split-fp32 tile against an fp64 evaluation of the same fp32 operands: within 1e-6 of each output's sum of
    |terms| (fp32 accumulation in another order: measured 2.6e-7 at worst; one dropped bf16 piece would be 4e-6 to 2e-3)."""
    xd = x.double()
    want = F.conv2d(xd, w_fold, bias.double().cpu(), 1, 1)
    mass = F.conv2d(xd.abs(), w_fold.abs(), bias.double().abs().cpu(), 1, 1)
    if res is not None:
        want, mass = want + res.double(), mass + res.double().abs()
    if act:
        want = F.relu(want)
    err = ((got_nhwc.double().cpu().permute(0, 3, 1, 2) - want).abs() / mass).max().item()
    assert err <= 1e-6, f"{what}: {err:.3e} of the sum of |terms|"
    # the yardstick: the SAME convolution on this library's direct fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, fp32 operands, fp32 accumulate)
    # against the same fp64 evaluation: the split-fp32 tile must be as close (both are fp32 accumulations of exact products, in different orders)
    if direct is not None:
        err32 = ((direct.double().cpu().permute(0, 3, 1, 2) - want).abs() / mass).max().item()
        assert err <= 2.0 * err32 + 5e-8, f"{what}: {err:.3e} vs {err32:.3e} for the direct fp32 kernel"
        _x3_check.pairs.append((err, err32))
    return err


_x3_check.pairs = []


def test_f32x3_fuzz_against_torch():
    """Seeded random 3x3 / stride-1 fp32 problems through the split-fp32 tile (csrc/igemm_f32x3_ws.hip): Cin multiples of 16, Cout
    multiples of 4 (ragged last channel slice), widths that are and are not multiples of 16 (closed-form and dealt-out pixel-to-column
    assignment), part rows / whole images / several images per tile, ragged last tiles, with and without residual / ReLU -- against an
    fp64 F.conv2d of the same fp32 operands with full-mantissa values.  The folded weights are read back from the packed pieces (which
    checks the pack: the three pieces must add up to the fp32 fold) and compared with the CPU's own BatchNorm fold."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20261001)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    worst = 0.0
    for case in range(30):
        ci = 16 * ri(1, 12)
        co = 4 * ri(1, 48)
        H, W, B = ri(1, 40), ri(1, 70), ri(1, 5)
        if case == 5:
            H, W, B = 8, 8, 9                     # several images per tile, ragged last tile, dealt-out columns
        if case == 6:
            H, W, B = 64, 64, 2                   # part rows of an image per tile
        if case == 7:
            H, W, B = 12, 9, 5                    # CPN's smallest map
        if case == 8:
            H, W, B, ci, co = 16, 16, 3, 128, 128
        act, res = ri(0, 1), bool(ri(0, 1))
        x = torch.randn(B, ci, H, W, generator=rng) * torch.rand(B, ci, H, W, generator=rng).pow(3)     # magnitudes over several binades
        w = torch.randn(co, ci, 3, 3, generator=rng) / (ci * 9) ** 0.5
        bnp = (torch.rand(co, generator=rng) + 0.5, torch.randn(co, generator=rng) * 0.1, torch.randn(co, generator=rng) * 0.1,
               torch.rand(co, generator=rng) * 0.4 + 0.8)
        wp, bias = capf.pack_conv_f32x3(w.cuda(), tuple(t.cuda() for t in bnp))
        w_fold = _x3_fold(wp, co, ci)
        sc = bnp[0].double() / torch.sqrt(bnp[3].double() + 1e-5)
        assert torch.allclose(w_fold, w.double() * sc.view(-1, 1, 1, 1), rtol=1e-6, atol=1e-9)
        assert torch.equal(w_fold.float().double(), w_fold)                    # an fp32 number: the pieces lose nothing
        r = torch.randn(B, co, H, W, generator=rng) if res else None
        got, = capf.conv_nhwc_f32x3_group([(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, act,
                                            r.permute(0, 2, 3, 1).contiguous().cuda() if res else None, co)])
        wd, bd = capf.pack_conv(w.cuda(), tuple(t.cuda() for t in bnp))
        direct = capf.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wd, bd, 3, 1, act, r.permute(0, 2, 3, 1).contiguous().cuda() if res else None)
        worst = max(worst, _x3_check(got, x, w_fold, bias, r, act, f"case {case}: Cin {ci} Cout {co} {H}x{W} B{B} act {act} res {res}", direct))
    e32 = max(b for _, b in _x3_check.pairs)
    print(f"split-fp32 tile, 30 random problems: worst |error| {worst:.2e} of the sum of |terms| (the direct fp32 MFMA kernel on the same problems: {e32:.2e})")


@pytest.mark.parametrize("chans,B", [((32, 64, 128, 256), 9), ((48, 96, 192, 384), 5), ((64, 128, 256, 512), 3)])
def test_grouped_f32x3_launch_matches_torch_and_single_launches(chans, B):
    """The four HRNet branch convs (3x3, stride 1, residual, ReLU) as ONE grouped launch of the split-fp32 tile -- problems of different
    K and tile geometry in one grid, padded tile ids -- against fp64 F.conv2d, and bit-identical to the four single launches."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(sum(chans) + B)
    probs, refs = [], []
    for i, c in enumerate(chans):
        r = 64 >> i
        x = torch.randn(B, c, r, r, generator=g)
        w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
        bnp = (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
               torch.rand(c, generator=g) * 0.4 + 0.8)
        res = torch.randn(B, c, r, r, generator=g)
        wp, bias = capf.pack_conv_f32x3(w.cuda(), tuple(t.cuda() for t in bnp))
        refs.append((x, _x3_fold(wp, c, c), bias, res))
        probs.append((x.permute(0, 2, 3, 1).contiguous().cuda(), wp, bias, 1, res.permute(0, 2, 3, 1).contiguous().cuda(), c))
    outs = capf.conv_nhwc_f32x3_group(probs)
    for y, (x, wf, bias, res), pr in zip(outs, refs, probs):
        _x3_check(y, x, wf, bias, res, 1, f"{x.shape}")
        single, = capf.conv_nhwc_f32x3_group([pr])
        assert torch.equal(single, y)


def test_f32x3_engine_path_agrees_with_the_winograd_kernels():
    """Batch 32 HRNet-32 fp32: the product plan runs the branch levels on igemm_f32x3_group_ws_kernel, a plan with CAPF_PLAN_NO_F32X3 on
    the Winograd kernels.  Same fp32 operands, two fp32-accurate evaluations: the context maps and the poses agree to fp32 roundoff
    accumulated over the backbone (the F(4,3) side contributes most of it)."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_NO_F32X3
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(32, 256, 256, seed=17)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps, outs, kernels = [], [], []
    for flags in (0, PLAN_NO_F32X3):
        with contextlib.redirect_stdout(io.StringIO()):
            model = CA_PF(cfg, compute_dtype="fp32", plan_flags=flags).eval()
        synth.load_synthetic(model, seed=4, bn_mode="random")
        model = model.cuda()
        with torch.no_grad():
            outs.append(model(img, k2d, kc.clone()).clone())
            eng = model.engine_for(img)
        kernels.append(set(k for _, k, _ in eng.op_table(32) if k))
        maps.append([eng.tensor(f"feat{l}").float().clone() for l in range(4)])
    assert any(k.startswith("igemm_f32h2_") for k in kernels[0]) and not any(k.startswith(("igemm_f32x3", "igemm_f32h2_")) for k in kernels[1])
    assert any(k.startswith("igemm_wino") for k in kernels[1])
    for a, b in zip(*maps):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"split-fp32 vs Winograd: relative L2 {rel:.2e}")
        assert rel < 2e-5
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-4 * outs[1].abs().max().item()


# ---- round 5: the default split-fp32 tile -- two block-scaled fp16 pieces per operand, three piece products (csrc/igemm_f32h2_ws.hip) ----

def _h2_fold(wp, co, ci):
    """Undo capf_op_pack_conv_f32h2: [32-channel slice][Cin / 16][piece 2][tap][32][quad position][8] fp16, then the fp32 inverse channel
    scales -> (folded weights [co, ci, 3, 3] in fp64 = (piece 0 + piece 1) / scale, the channel scales)."""
    nsl = (co + 31) // 32
    npieces = nsl * (ci // 16) * 2 * 9 * 32 * 16
    raw = wp.cpu()
    winv = raw[npieces:npieces + 2 * nsl * 32].view(torch.float32).double()
    t = raw[:npieces].view(torch.float16).double().view(nsl, ci // 16, 2, 9, 32, 2, 8).sum(dim=2)
    n = torch.arange(32)
    swap = ((n >> 3) & 1).bool()
    t = torch.where(swap[None, None, None, :, None, None], t.flip(4), t)           # quad position -> channel half
    w = t.permute(0, 3, 1, 4, 5, 2).reshape(nsl * 32, ci, 9) * winv.view(-1, 1, 1)
    return w[:co].reshape(co, ci, 3, 3).contiguous(), winv[:co]


def _h2_problem(rng, ci, co, H, W, B, act, res, xscale=None, gamma_spread=0):
    from capf import lib as capf
    x = torch.randn(B, ci, H, W, generator=rng) * torch.rand(B, ci, H, W, generator=rng).pow(3)     # magnitudes over several binades
    if xscale is not None:
        x = x * xscale
    w = torch.randn(co, ci, 3, 3, generator=rng) / (ci * 9) ** 0.5
    gam = torch.rand(co, generator=rng) + 0.5
    if gamma_spread:
        gam = gam * torch.exp2(torch.randint(-gamma_spread, gamma_spread + 1, (co,), generator=rng).float())
    bnp = (gam, torch.randn(co, generator=rng) * 0.1, torch.randn(co, generator=rng) * 0.1, torch.rand(co, generator=rng) * 0.4 + 0.8)
    bn_cuda = tuple(t.cuda() for t in bnp)
    wp, bias = capf.pack_conv_f32h2(w.cuda(), bn_cuda)
    wp3, bias3 = capf.pack_conv_f32x3(w.cuda(), bn_cuda)
    w_fold = _x3_fold(wp3, co, ci)                                    # the exact fp32 fold (the three-piece pack loses nothing: tested above)
    assert torch.equal(bias, bias3)
    w_h2, winv = _h2_fold(wp, co, ci)
    # the pack: two fp16 pieces of (fold * channel scale) reproduce the fold to 2^-23 (denormal second pieces: 2^-39 of the channel's largest)
    cmax = w_fold.abs().amax(dim=(1, 2, 3), keepdim=True)
    assert ((w_h2 - w_fold).abs() <= 2.0 ** -23 * w_fold.abs() + 2.0 ** -38 * cmax).all()
    sc = torch.log2(winv)
    assert torch.equal(sc, sc.round())                                # powers of two ...
    top = cmax.flatten() / winv
    assert ((top >= 2.0 ** 14) & (top < 2.0 ** 15) | (cmax.flatten() == 0)).all()        # ... that put the channel's largest weight in [2^14, 2^15)
    r = torch.randn(B, co, H, W, generator=rng) if res else None
    if r is not None and xscale is not None:
        r = r * float(xscale if not torch.is_tensor(xscale) else xscale.abs().max())
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    rg = r.permute(0, 2, 3, 1).contiguous().cuda() if res else None
    return x, w, bnp, wp, bias, w_fold, r, xg, rg


def test_f32h2_fuzz_against_torch():
    """The 30 seeded problems of test_f32x3_fuzz_against_torch through the DEFAULT split-fp32 tile (two fp16 pieces under exact power-of-two
    block scales, three piece products): same bound against the fp64 F.conv2d of the fp32 operands -- 1e-6 of each output's sum of |terms| --
    and the same yardstick, this library's direct fp32 MFMA kernel on the same problem (<= 2 x its error).  The pack is read back: the two
    pieces of a weight, unscaled, are within 2^-23 of the fp32 fold; the scales are powers of two."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20261001)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    worst, n0 = 0.0, len(_x3_check.pairs)
    for case in range(30):
        ci = 16 * ri(1, 12)
        co = 4 * ri(1, 48)
        H, W, B = ri(1, 40), ri(1, 70), ri(1, 5)
        if case == 5:
            H, W, B = 8, 8, 9
        if case == 6:
            H, W, B = 64, 64, 2
        if case == 7:
            H, W, B = 12, 9, 5
        if case == 8:
            H, W, B, ci, co = 16, 16, 3, 128, 128
        act, res = ri(0, 1), bool(ri(0, 1))
        x, w, bnp, wp, bias, w_fold, r, xg, rg = _h2_problem(rng, ci, co, H, W, B, act, res)
        got, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, act, rg, co)])
        wd, bd = capf.pack_conv(w.cuda(), tuple(t.cuda() for t in bnp))
        direct = capf.conv_nhwc(xg, wd, bd, 3, 1, act, rg)
        worst = max(worst, _x3_check(got, x, w_fold, bias, r, act, f"h2 case {case}: Cin {ci} Cout {co} {H}x{W} B{B} act {act} res {res}", direct))
    e32 = max(b for _, b in _x3_check.pairs[n0:])
    print(f"two-fp16-piece tile, 30 random problems: worst |error| {worst:.2e} of the sum of |terms| (the direct fp32 MFMA kernel on the same problems: {e32:.2e})")


@pytest.mark.parametrize("mode", ["regions", "huge", "tiny", "channels", "dead"])
def test_f32h2_block_scales_follow_the_data(mode):
    """What the block scales are for: activations whose magnitude changes by 2^+-12 from one image region / channel group to the next,
    tensors around 1e20 and 1e-20 (far outside fp16's range without the scale), BatchNorm folds whose channels differ by 2^+-10, and
    tiles that are entirely zero (dead ReLU regions: the scale clamps, the bias must survive) -- same bound as the fuzz test."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed({"regions": 1, "huge": 2, "tiny": 3, "channels": 4, "dead": 5}[mode])
    ci, co, H, W, B = 64, 96, 32, 32, 3
    xscale, spread = None, 0
    if mode == "regions":
        xscale = torch.exp2(torch.randint(-12, 13, (B, ci // 16, H // 8, 1), generator=rng).float()).repeat_interleave(16, 1).repeat_interleave(8, 2)
    elif mode == "huge":
        xscale = 1e20
    elif mode == "tiny":
        xscale = 1e-20
    elif mode == "channels":
        spread = 10
    elif mode == "dead":
        xscale = torch.ones(B, 1, H, 1)
        xscale[1] = 0.0                                              # a whole image of zeros
        xscale[2, :, :16] = 0.0                                      # and half of another
    x, w, bnp, wp, bias, w_fold, r, xg, rg = _h2_problem(rng, ci, co, H, W, B, 0, True, xscale, spread)
    got, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, 0, rg, co)])
    assert torch.isfinite(got).all()
    err = _x3_check(got, x, w_fold, bias, r, 0, f"h2 {mode}")
    print(f"two-fp16-piece tile, {mode}: {err:.2e} of the sum of |terms|")


def _flat_with_outlier(g, shape, kind, e, where):
    """values of magnitude [0.5, 1) with random signs -- a FLAT block -- and one outlier 2^e above them: `pixel` = every channel of one
    position, `element` = one value, `channel` = one channel over the whole first sample.  shape = (B, C, ...)."""
    x = (torch.rand(shape, generator=g) * 0.5 + 0.5) * (torch.randint(0, 2, shape, generator=g).float() * 2 - 1)
    if kind == "pixel":
        x[(0, slice(None)) + where] *= 2.0 ** e
    elif kind == "element":
        x[(0, 3) + where] *= 2.0 ** e
    elif kind == "channel":
        x[0, 20] *= 2.0 ** e
    return x


def _range_check(got, direct, want, mass, slack, what, clean, base=1e-6):
    """One kernel of the two-piece family on a flat tensor with an outlier, per output:
      * |got - fp64| <= 2.5 base sum|terms| + 2^-38 slack (include/capf.h, THE BOUND: the first term is what an fp32 accumulation costs when
        ONE term dominates the sum -- this library's fp32-pipe kernel on the same operands, `direct`, reaches 1 - 2 base there too; the second
        is the block-scale term, slack = sum_c M_c W_c);
      * the yardstick of every other test: at most 2 x the fp32-pipe kernel's error (+ the block-scale term);
      * where `clean` (blocks that hold no outlier): the plain `base` of the fuzz tests.
    Returns (worst error / sum|terms| over the clean outputs, over the rest, the fp32-pipe kernel's worst)."""
    rel = (got - want).abs() / mass
    rel32 = ((direct - want).abs() / mass).max().item()
    block_term = 2.0 ** -38 * slack / mass
    assert (rel <= 2.5 * base + block_term).all(), f"{what}: {(rel / (2.5 * base + block_term)).max().item():.2f} of the promised bound"
    assert (rel <= 2.0 * rel32 + 5e-8 + block_term).all(), f"{what}: {rel.max().item():.3e} vs {rel32:.3e} for the fp32-pipe kernel"
    worst_clean = rel[clean].max().item() if clean.any() else 0.0
    assert worst_clean <= base, f"{what}: {worst_clean:.3e} of the sum of |terms| in blocks without an outlier"
    return worst_clean, rel[~clean].max().item() if (~clean).any() else 0.0, rel32


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pixel", "element", "channel"])
@pytest.mark.parametrize("e", [8, 12, 16, 20])
def test_f32h2_dynamic_range_inside_a_block(kind, e):
    """the two-piece arithmetic's error is relative to the BLOCK maximum, and no test had a wide dynamic range INSIDE one block
    (test_f32h2_block_scales_follow_the_data steps its magnitudes on the kernel's own block / chunk grid).  Here a flat tensor carries one
    outlier 2^8 .. 2^20 above everything else -- one pixel, one element, one whole channel of the first image -- for the conv tile
    (igemm_f32h2_ws), the two-piece GEMM as a conv and as a linear (igemm_f32h2g) and the weight gradient (wgrad_tn_h2).  Asserted per output
    (_range_check): the bound include/capf.h promises with M_c taken over the outlier's whole image (an over-estimate of the block), the
    fp32-pipe kernel on the same operands as the yardstick, and the plain fuzz-test bound for every output whose blocks hold no outlier (the
    other images / row blocks / column tiles)."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(1000 * e + len(kind))
    B, ci, co, H, W = 3, 64, 64, 32, 32
    # ---- conv tile: 256-pixel tiles are 8 rows of one image; chunks of 16 channels
    x = _flat_with_outlier(g, (B, ci, H, W), kind, e, (5, 7))
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    bnp = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) * 0.4 + 0.8)
    bn_cuda = tuple(t.cuda() for t in bnp)
    wp, bias = capf.pack_conv_f32h2(w.cuda(), bn_cuda)
    wd, bd = capf.pack_conv(w.cuda(), bn_cuda)
    w_fold = wd.cpu().double()[:, :9 * ci].view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
    r = torch.randn(B, co, H, W, generator=g)
    xg, rg = x.permute(0, 2, 3, 1).contiguous().cuda(), r.permute(0, 2, 3, 1).contiguous().cuda()
    got, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, 0, rg, co)])
    direct = capf.conv_nhwc(xg, wd, bd, 3, 1, 0, rg)
    assert torch.isfinite(got).all()
    xd = x.double()
    want = F.conv2d(xd, w_fold, bias.double().cpu(), 1, 1) + r.double()
    mass = F.conv2d(xd.abs(), w_fold.abs(), bias.double().abs().cpu(), 1, 1) + r.double().abs()
    m_c = xd.abs().view(B, ci // 16, 16, H, W).amax(dim=(2, 3, 4))                      # [B, chunk]: largest value of the image's chunk
    w_c = w_fold.abs().view(co, ci // 16, 16 * 9).sum(dim=2)                            # [co, chunk]
    slack = (m_c @ w_c.t())[:, :, None, None].expand_as(want)
    clean = torch.zeros_like(want, dtype=torch.bool)
    clean[1:] = True
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)
    c1, d1, f1 = _range_check(nchw(got), nchw(direct), want, mass, slack, f"conv tile, {kind} 2^{e}", clean)
    # ---- the two-piece GEMM as a stride-2 conv (chunks of 32 of k = (kh, kw, ci): one half of the channels of one tap)
    wpg, biasg = capf.pack_f32h2_gemm(w.cuda(), bn_cuda)
    got2 = capf.conv_nhwc_f32h2g(xg, wpg, biasg, 3, 2, 0, None, co)
    direct2 = capf.conv_nhwc(xg, wd, bd, 3, 2, 0, None)
    want2 = F.conv2d(xd, w_fold, biasg.double().cpu(), 2, 1)
    mass2 = F.conv2d(xd.abs(), w_fold.abs(), biasg.double().abs().cpu(), 2, 1)
    m_h = xd.abs().view(B, ci // 32, 32, H, W).amax(dim=(2, 3, 4))
    w_h = w_fold.abs().view(co, ci // 32, 32 * 9).sum(dim=2)
    slack2 = (m_h @ w_h.t())[:, :, None, None].expand_as(want2)
    clean2 = torch.zeros_like(want2, dtype=torch.bool)
    clean2[1:] = True                                                                   # (16 x 16 outputs per image: row blocks of 32 never span images)
    c2, d2, f2 = _range_check(nchw(got2), nchw(direct2), want2, mass2, slack2, f"two-piece GEMM (conv), {kind} 2^{e}", clean2)
    # ---- ... as a linear (the lifter's joint-block shape at batch 64): row blocks of 32, chunks of 32 columns -- M_c exactly
    M, K, N = 1088, 640, 640
    xl = _flat_with_outlier(g, (1, M, K), "none", 0, ())[0]
    if kind == "pixel":
        xl[100, :] *= 2.0 ** e                                                          # a whole row
    elif kind == "element":
        xl[100, 7] *= 2.0 ** e
    else:
        xl[:, 7] *= 2.0 ** e                                                            # a whole column: every row block holds the outlier
    wl = torch.randn(N, K, generator=g) / K ** 0.5
    bl = torch.randn(N, generator=g) * 0.1
    wpl, _ = capf.pack_f32h2_gemm(wl.cuda())
    gotl = capf.linear_f32h2g(xl.cuda(), wpl, bl.cuda(), N, 0, None).cpu().double()
    directl = capf.linear(xl.cuda(), wl.cuda(), bl.cuda(), 0, None).cpu().double()
    wantl = xl.double() @ wl.double().t() + bl.double()
    massl = xl.double().abs() @ wl.double().abs().t() + bl.double().abs()
    m_rc = xl.double().abs().view(M // 32, 32, K // 32, 32).amax(dim=(1, 3))               # [row block, chunk]
    w_nc = wl.double().abs().view(N, K // 32, 32).sum(dim=2)                            # [N, chunk]
    slackl = (m_rc @ w_nc.t()).repeat_interleave(32, 0)
    cleanl = (m_rc.amax(dim=1) < 2.0).repeat_interleave(32, 0)[:, None].expand_as(wantl)
    c3, d3, f3 = _range_check(gotl, directl, wantl, massl, slackl, f"two-piece GEMM (linear), {kind} 2^{e}", cleanl, 1.2e-6)
    # ---- the weight gradient: both operands split, scales that only go down along M, so everything BEHIND the outlier runs on its scale
    # (a 2048-term fp32 sum: base 2e-6, what the fp32-pipe kernel is held to in the training tests)
    Mw, Nw, Kw = 2048, 256, 128
    dy = _flat_with_outlier(g, (1, Mw, Nw), "none", 0, ())[0]
    if kind == "pixel":
        dy[300, :128] *= 2.0 ** e                                                       # a whole row of the first 128-column tile
    else:
        dy[300, 5 if kind == "element" else 70] *= 2.0 ** e                             # one value (the second 128-column tile stays flat)
    xw = _flat_with_outlier(g, (1, Mw, Kw), "none", 0, ())[0]
    gw, gb = capf.wgrad(dy.cuda(), xw.cuda(), True)
    gw32, gb32 = capf.wgrad(dy.cuda(), xw.cuda(), False)
    wantw = dy.double().t() @ xw.double()
    massw = dy.double().abs().t() @ xw.double().abs()
    blk_max = dy.double().abs().view(Mw, Nw // 128, 128).amax(dim=(0, 2))                 # largest dY value a 128-column tile ever sees
    slackw = (blk_max.repeat_interleave(128)[:, None] * xw.double().abs().sum(dim=0)[None, :]
              + xw.double().abs().max() * dy.double().abs().sum(dim=0)[:, None])
    cleanw = (blk_max.repeat_interleave(128) < 2.0)[:, None].expand_as(wantw)
    c4, d4, f4 = _range_check(gw.cpu().double(), gw32.cpu().double(), wantw, massw, slackw, f"weight gradient, {kind} 2^{e}", cleanw, 2e-6)
    col_mass = dy.double().abs().sum(dim=0)
    for b_ in (gb, gb32):                                                               # (the bias column sums: plain fp32 sums in both kernels)
        assert ((b_.cpu().double() - dy.double().sum(dim=0)).abs() <= 1e-5 * col_mass).all()
    print(f"outlier 2^{e} ({kind}): error / sum|terms| in blocks without | with the outlier [fp32-pipe kernel, same operands] -- conv tile {c1:.1e} | {d1:.1e} [{f1:.1e}];"
          f" GEMM conv {c2:.1e} | {d2:.1e} [{f2:.1e}]; GEMM linear {c3:.1e} | {d3:.1e} [{f3:.1e}]; weight gradient {c4:.1e} | {d4:.1e} [{f4:.1e}]")


@pytest.mark.gpu
def test_f32h2_scales_do_not_chase_a_chunk_of_zeros_into_overflow():
    """an all-zero chunk (padding taps, dead ReLU channels, DropPath rows) used to pull the running scale to its ceiling; behind a
    chunk whose maximum is ~1e14 the accumulators (2^35 in units of that chunk's scale) were multiplied by 2^90 and overflowed.  Channels 0-31 at
    1e14, channels 32-63 zero, for the conv tile and the GEMM: finite and within the usual bound."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(99)
    B, ci, co, H, W = 2, 64, 64, 16, 16
    x = torch.randn(B, ci, H, W, generator=g) * 1e14
    x[:, 32:] = 0.0
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    wp, bias = capf.pack_conv_f32h2(w.cuda(), None)
    w_fold = _x3_fold(capf.pack_conv_f32x3(w.cuda(), None)[0], co, ci)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    got, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, 0, None, co)])
    assert torch.isfinite(got).all()
    _x3_check(got, x, w_fold, bias, None, 0, "conv tile behind a chunk of zeros")
    wpg, biasg = capf.pack_f32h2_gemm(w.cuda(), None)
    got2 = capf.conv_nhwc_f32h2g(xg, wpg, biasg, 3, 1, 0, None, co)
    assert torch.isfinite(got2).all()
    _x3_check(got2, x, w_fold, biasg, None, 0, "two-piece GEMM behind a chunk of zeros")
    xl = torch.randn(256, 128, generator=g) * 1e14
    xl[:, 64:] = 0.0
    wl = torch.randn(128, 128, generator=g) / 128 ** 0.5
    wpl, _ = capf.pack_f32h2_gemm(wl.cuda())
    gotl = capf.linear_f32h2g(xl.cuda(), wpl, torch.zeros(128).cuda(), 128, 0, None).cpu().double()
    wantl = xl.double() @ wl.double().t()
    assert torch.isfinite(gotl).all()
    assert ((gotl - wantl).abs() / (xl.double().abs() @ wl.double().abs().t())).max().item() <= 1e-6


@pytest.mark.parametrize("chans,B", [((32, 64, 128, 256), 9), ((48, 96, 192, 384), 5), ((64, 128, 256, 512), 3)])
def test_grouped_f32h2_launch_matches_torch_and_single_launches(chans, B):
    """The four HRNet branch convs as ONE grouped launch of the two-fp16-piece tile against fp64 F.conv2d, bit-identical to the four
    single launches."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(sum(chans) + B)
    probs, refs = [], []
    for i, c in enumerate(chans):
        r_ = 64 >> i
        x, w, bnp, wp, bias, w_fold, r, xg, rg = _h2_problem(g, c, c, r_, r_, B, 1, True)
        refs.append((x, w_fold, bias, r))
        probs.append((xg, wp, bias, 1, rg, c))
    outs = capf.conv_nhwc_f32h2_group(probs)
    for y, (x, wf, bias, res), pr in zip(outs, refs, probs):
        _x3_check(y, x, wf, bias, res, 1, f"h2 {x.shape}")
        single, = capf.conv_nhwc_f32h2_group([pr])
        assert torch.equal(single, y)


def test_f32h2_wide_and_narrow_tiles_compute_the_same_bits():
    """From 512 tiles of 64 channels a conv runs 64-channel tiles (two blocks per CU), below 32-channel ones (three): the K order and the
    block scales do not depend on the width, so the first images of a 256-image batch (64-channel tiles) must equal the same images run
    as a batch of 3 (32-channel tiles) bit for bit -- at 16x16 a tile is one image."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(77)
    c, B = 128, 256
    x, w, bnp, wp, bias, w_fold, r, xg, rg = _h2_problem(g, c, c, 16, 16, B, 1, True)
    big, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, 1, rg, c)])
    small, = capf.conv_nhwc_f32h2_group([(xg[:3].contiguous(), wp, bias, 1, rg[:3].contiguous(), c)])
    assert torch.equal(big[:3], small)
    _x3_check(big[:8], x[:8], w_fold, bias, r[:8], 1, "h2 64-channel tiles")


def test_f32h2_engine_path_agrees_with_the_exact_three_piece_plan():
    """Batch 32 HRNet-32 fp32: the product plan (two-fp16-piece tile) against CAPF_PLAN_F32X3_EXACT (round 4's exact-operand tile) and
    CAPF_PLAN_NO_F32X3 (fp32 matrix pipe): the context maps of the two split plans agree to fp32 roundoff -- closer to each other than
    either is to the Winograd plan."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_F32X3_EXACT, PLAN_NO_F32X3
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(32, 256, 256, seed=17)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps, outs, kernels = [], [], []
    for flags in (0, PLAN_F32X3_EXACT, PLAN_NO_F32X3):
        with contextlib.redirect_stdout(io.StringIO()):
            model = CA_PF(cfg, compute_dtype="fp32", plan_flags=flags).eval()
        synth.load_synthetic(model, seed=4, bn_mode="random")
        model = model.cuda()
        with torch.no_grad():
            outs.append(model(img, k2d, kc.clone()).clone())
            eng = model.engine_for(img)
        kernels.append(set(k for _, k, _ in eng.op_table(32) if k))
        maps.append([eng.tensor(f"feat{l}").float().clone() for l in range(4)])
    assert any(k.startswith("igemm_f32h2_") for k in kernels[0]) and not any(k.startswith("igemm_f32x3") for k in kernels[0])
    assert any(k.startswith("igemm_f32x3") for k in kernels[1]) and not any(k.startswith("igemm_f32h2_") for k in kernels[1])
    for a, b, c in zip(*maps):
        rel, rel_w = ((a - b).norm() / b.norm()).item(), ((a - c).norm() / c.norm()).item()
        print(f"two-piece vs exact three-piece plan: relative L2 {rel:.2e}   (vs the Winograd plan {rel_w:.2e})")
        assert rel < 2e-6 and rel_w < 2e-5
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * outs[1].abs().max().item()


# ---- round 5: the two-fp16-piece arithmetic for every other fp32 conv / linear (csrc/igemm_f32h2.hip) ----

def _h2g_unpack(wp, n, k):
    """Undo capf_op_pack_f32h2_gemm: [N][Kpad / 32][piece 2][32] fp16 + [N] fp32 inverse scales -> (weights [N, K] fp64, scales [N])."""
    kpad = (k + 31) // 32 * 32
    raw = wp.cpu()
    winv = raw[n * kpad:n * kpad + n].double()
    t = raw[:n * kpad].view(torch.float16).double().view(n, kpad // 32, 2, 32).sum(dim=2).reshape(n, kpad)
    return (t * winv.view(-1, 1))[:, :k], winv


def _h2g_conv_check(got_nhwc, x, w_fold, bias, res, act, stride, what, direct):
    ks = w_fold.shape[-1]
    xd = x.double()
    want = F.conv2d(xd, w_fold, bias.double().cpu(), stride, ks // 2)
    mass = F.conv2d(xd.abs(), w_fold.abs(), bias.double().abs().cpu(), stride, ks // 2)
    if res is not None:
        want, mass = want + res.double(), mass + res.double().abs()
    if act:
        want = F.relu(want)
    err = ((got_nhwc.double().cpu().permute(0, 3, 1, 2) - want).abs() / mass).max().item()
    err32 = ((direct.double().cpu().permute(0, 3, 1, 2) - want).abs() / mass).max().item()
    assert err <= 1e-6, f"{what}: {err:.3e} of the sum of |terms|"
    assert err <= 2.0 * err32 + 5e-8, f"{what}: {err:.3e} vs {err32:.3e} for the direct fp32 kernel"
    return err, err32


def test_f32h2g_conv_fuzz_against_torch():
    """1x1 / 3x3 / 5x5 convs, stride 1 and 2, channel counts that are and are not multiples of 32 (block-uniform and per-thread tap walk),
    ragged M and N tiles, with and without residual / ReLU, through the two-fp16-piece GEMM (csrc/igemm_f32h2.hip) against an fp64 F.conv2d
    of the same fp32 operands -- 1e-6 of the sum of |terms|, and at most 2 x the error of this library's direct fp32 MFMA kernel on the same
    problem.  The pack is read back: (piece 0 + piece 1) / scale is within 2^-23 of the fp32 fold the fp32 pack holds."""
    from capf import lib as capf
    rng = torch.Generator().manual_seed(20261002)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    worst, worst32 = 0.0, 0.0
    for case in range(24):
        ks = (1, 3, 3, 5)[ri(0, 3)]
        stride = ri(1, 2)
        ci = 4 * ri(1, 40) if case % 3 else 32 * ri(1, 8)
        co = 4 * ri(1, 70)
        H, W, B = ri(3, 36), ri(3, 40), ri(1, 6)
        act, res = ri(0, 1), bool(ri(0, 1))
        x = torch.randn(B, ci, H, W, generator=rng) * torch.rand(B, ci, H, W, generator=rng).pow(3)
        w = torch.randn(co, ci, ks, ks, generator=rng) / (ci * ks * ks) ** 0.5
        bnp = (torch.rand(co, generator=rng) + 0.5, torch.randn(co, generator=rng) * 0.1, torch.randn(co, generator=rng) * 0.1,
               torch.rand(co, generator=rng) * 0.4 + 0.8)
        bn_cuda = tuple(t.cuda() for t in bnp)
        wd, bd = capf.pack_conv(w.cuda(), bn_cuda)                                   # the fp32 pack: [Cout][Kpad], k = (kh, kw, ci)
        k = ks * ks * ci
        w_fold = wd.cpu().double()[:, :k].view(co, ks, ks, ci).permute(0, 3, 1, 2).contiguous()
        wp, bias = capf.pack_f32h2_gemm(w.cuda(), bn_cuda)
        assert torch.equal(bias, bd)
        w_h2, winv = _h2g_unpack(wp, co, k)
        flat = wd.cpu().double()[:, :k]
        cmax = flat.abs().amax(dim=1, keepdim=True)
        assert ((w_h2 - flat).abs() <= 2.0 ** -23 * flat.abs() + 2.0 ** -38 * cmax).all()
        assert torch.equal(torch.log2(winv), torch.log2(winv).round())
        pad = ks // 2
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        r = torch.randn(B, co, Ho, Wo, generator=rng) if res else None
        xg = x.permute(0, 2, 3, 1).contiguous().cuda()
        rg = r.permute(0, 2, 3, 1).contiguous().cuda() if res else None
        got = capf.conv_nhwc_f32h2g(xg, wp, bias, ks, stride, act, rg, co)
        direct = capf.conv_nhwc(xg, wd, bd, ks, stride, act, rg)
        e, e32 = _h2g_conv_check(got, x, w_fold, bias, r, act, stride, f"h2g conv case {case}: {ci}->{co} k{ks} s{stride} {H}x{W} B{B} act {act} res {res}", direct)
        worst, worst32 = max(worst, e), max(worst32, e32)
    print(f"two-fp16-piece GEMM, 24 random convs: worst |error| {worst:.2e} of the sum of |terms| (the direct fp32 MFMA kernel: {worst32:.2e})")


def test_f32h2g_grouped_convs_match_single_launches():
    """The convs of an HRNet fuse layer (1x1 at the low resolutions, 3x3 stride-2 chains) as ONE grid of igemm_f32h2g_group_kernel: bit-identical to
    the single launches, and against fp64."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(99)
    B = 7
    shapes = [(64, 32, 32, 1, 1), (128, 32, 16, 1, 1), (32, 64, 64, 3, 2), (64, 128, 32, 3, 2), (256, 64, 8, 1, 1), (128, 256, 16, 3, 2)]      # (Cin, Cout, HW, ks, stride)
    probs, refs = [], []
    for ci, co, hw, ks, st in shapes:
        x = torch.randn(B, ci, hw, hw, generator=g)
        w = torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5
        bnp = tuple(t.cuda() for t in (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
                                       torch.rand(co, generator=g) * 0.4 + 0.8))
        wd, bd = capf.pack_conv(w.cuda(), bnp)
        wp, bias = capf.pack_f32h2_gemm(w.cuda(), bnp)
        xg = x.permute(0, 2, 3, 1).contiguous().cuda()
        probs.append((xg, wp, bias, ks, st, 0, None, co))
        refs.append((x, wd, bd, ks, st, ci, co))
    outs = capf.conv_nhwc_f32h2g_group(probs)
    for y, pr, (x, wd, bd, ks, st, ci, co) in zip(outs, probs, refs):
        single = capf.conv_nhwc_f32h2g(pr[0], pr[1], pr[2], ks, st, 0, None, co)
        assert torch.equal(single, y)
        w_fold = wd.cpu().double()[:, :ks * ks * ci].view(co, ks, ks, ci).permute(0, 3, 1, 2).contiguous()
        _h2g_conv_check(y, x, w_fold, bd, None, 0, st, f"grouped h2g {ci}->{co}", capf.conv_nhwc(pr[0], wd, bd, ks, st, 0, None))


@pytest.mark.parametrize("M,K,N,act,res", [(1088, 640, 1920, 0, False), (1088, 640, 640, 0, True), (1088, 640, 1280, 2, False), (1088, 1280, 640, 0, True),
                                           (5440, 128, 128, 0, True), (5440, 256, 128, 0, True), (37, 96, 52, 2, True),
                                           (8704, 640, 1920, 0, False), (8650, 256, 1920, 2, True)])     # (the 128 x 128 tile: joint qkv at batch 512; ragged rows)
def test_f32h2g_linear_matches_fp64(M, K, N, act, res):
    """The lifter's projections at batch 64 (joint blocks: 17 B rows of 640; res blocks: 85 B rows of 128; pose_dformer.py:15-59) and a ragged
    one: y = act(x W^T + b (+ residual)) on the two-fp16-piece GEMM against fp64 (1e-6 of the sum of |terms|, GELU through its Lipschitz
    bound) and against this library's fp32 MFMA GEMM on the same problem."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g) * torch.rand(M, K, generator=g).pow(2)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    wp, _ = capf.pack_f32h2_gemm(w.cuda())
    w_h2, winv = _h2g_unpack(wp, N, K)
    cmax = w.double().abs().amax(dim=1, keepdim=True)
    assert ((w_h2 - w.double()).abs() <= 2.0 ** -23 * w.double().abs() + 2.0 ** -38 * cmax).all()
    got = capf.linear_f32h2g(x.cuda(), wp, b.cuda(), N, act, r.cuda() if res else None).cpu().double()
    direct = capf.linear(x.cuda(), w.cuda(), b.cuda(), act, r.cuda() if res else None).cpu().double()
    pre = x.double() @ w.double().t() + b.double()
    mass = x.double().abs() @ w.double().abs().t() + b.double().abs()
    if res:
        pre, mass = pre + r.double(), mass + r.double().abs()            # (the residual goes in before the activation, as in igemm_f32.hip)
    want = F.gelu(pre) if act == 2 else pre
    err, err32 = ((got - want).abs() / mass).max().item(), ((direct - want).abs() / mass).max().item()
    print(f"h2g linear {M}x{K}->{N} act {act} res {res}: {err:.2e} of the sum of |terms| (fp32 MFMA GEMM: {err32:.2e})")
    assert err <= 1.2e-6 and err <= 2.0 * err32 + 5e-8


@pytest.mark.parametrize("M,K,N,act,res,ragged", [(5440, 128, 384, 0, False, False), (5440, 128, 256, 2, False, False), (4352, 128, 256, 2, False, False),
                                                  (43520, 128, 384, 0, False, False), (201, 96, 52, 0, True, True), (1000, 256, 128, 2, True, False)])
def test_f32h2g_layernorm_fold_matches_fp64(M, K, N, act, res, ragged):
    """LayerNorm folded in front of a projection on the two-fp16-piece GEMM (res blocks' norm1 -> qkv and norm2 -> fc1 at batch 64 / 512, the
    context blocks' norm2 -> fc1; pose_dformer.py:62-79, 137-138; and a ragged problem): rows with offsets and scales far from a standard
    normal, so that a LayerNorm computed in the wrong place or order would show.  Against fp64 of LayerNorm + Linear (+ GELU), error in units of
    sum |x_hat| |w| + |b|: 2e-6 (the fp32 LayerNorm itself costs eps_fp32 * |mean| / sigma of x_hat: rows here have |mean| / sigma up to ~8;
    with offsets of 30 sigma the same kernel measures 3.6e-6, which is the LayerNorm's cancellation, not the split)."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(M + 3 * K + N)
    x = torch.randn(M, K, generator=g) * (torch.rand(M, 1, generator=g) * 2 + 0.5) + torch.randn(M, 1, generator=g)
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    eps = 1e-6
    wp, _ = capf.pack_f32h2_gemm(w.cuda())
    got = capf.linear_ln_f32h2g(x.cuda(), gamma.cuda(), beta.cuda(), eps, wp, b.cuda(), N, act, r.cuda() if res else None).cpu().double()
    xd = x.double()
    xh = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + eps) * gamma.double() + beta.double()
    pre = xh @ w.double().t() + b.double()
    mass = xh.abs() @ w.double().abs().t() + b.double().abs()
    if res:
        pre, mass = pre + r.double(), mass + r.double().abs()
    want = F.gelu(pre) if act == 2 else pre
    err = ((got - want).abs() / mass).max().item()
    print(f"h2g LayerNorm + linear {M}x{K}->{N} act {act} res {res}: {err:.2e} of the sum of |terms|")
    assert err <= 2e-6


def test_f32h2g_engine_path_agrees_with_the_fp32_pipe_gemms():
    """Batch 32 HRNet-32 fp32: the product plan runs the fuse / transition / lone convs and the lifter's projections on igemm_f32h2g, a plan
    with CAPF_PLAN_NO_F32H2_GEMM on igemm_f32 (fp32 matrix pipe).  Context maps and poses agree to fp32 roundoff."""
    import copy, contextlib, io
    from capf import synth
    from capf.lib import PLAN_NO_F32H2_GEMM
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    img, k2d, kc = synth.synth_inputs(32, 256, 256, seed=19)
    img, k2d, kc = img.cuda(), k2d.cuda(), kc.cuda()
    maps, outs, kernels = [], [], []
    for flags in (0, PLAN_NO_F32H2_GEMM):
        with contextlib.redirect_stdout(io.StringIO()):
            model = CA_PF(cfg, compute_dtype="fp32", plan_flags=flags).eval()
        synth.load_synthetic(model, seed=5, bn_mode="random")
        model = model.cuda()
        with torch.no_grad():
            outs.append(model(img, k2d, kc.clone()).clone())
            eng = model.engine_for(img)
        kernels.append([k for _, k, _ in eng.op_table(32) if k])
        maps.append([eng.tensor(f"feat{l}").float().clone() for l in range(4)])
    assert sum(k.startswith("igemm_f32h2g") for k in kernels[0]) >= 60 and not any(k.startswith("igemm_f32h2g") for k in kernels[1])
    assert any(k.startswith("igemm_f32<") for k in kernels[1])
    for a, b in zip(*maps):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"two-fp16-piece GEMM plan vs fp32-pipe GEMM plan: relative L2 {rel:.2e}")
        assert rel < 2e-6
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * outs[1].abs().max().item()


def test_split_fp32_kernels_take_tensors_beyond_2_gib():
    """The training bench's batch (512 frames) makes three backbone tensors larger than 2 GiB in fp32 (conv2's input, layer1's output into
    both transition1 convs).  The two-fp16-piece tile and the two-piece GEMM address from per-tile bases, so they take them -- checked
    bit for bit against the same frames run as two half batches (a tile of these maps never spans two frames' worth of scale blocks:
    64 x 64 maps give one tile per four rows; the GEMM's 128-pixel tiles split 4096-pixel frames evenly), and against fp64 on a few frames."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(2026)
    # transition1.0.0: 256 -> 32, 3x3 stride 1 at 64x64 (pose_hrnet.py:331-356), 544 frames: 2.28e9 bytes of input
    B, ci, co, hw = 544, 256, 32, 64
    x8, w, bnp, wp, bias, w_fold, _, xg8, _ = _h2_problem(g, ci, co, hw, hw, 8, 1, False)
    xg = xg8.repeat(B // 8, 1, 1, 1)                                     # [B, 64, 64, 256] fp32 NHWC
    xg += torch.arange(B, device="cuda").view(B, 1, 1, 1) * 1e-3          # (every frame different)
    assert xg.numel() * 4 > 2 ** 31
    big, = capf.conv_nhwc_f32h2_group([(xg, wp, bias, 1, None, co)])
    for lo in (0, B // 2):
        half, = capf.conv_nhwc_f32h2_group([(xg[lo:lo + B // 2].contiguous(), wp, bias, 1, None, co)])
        assert torch.equal(big[lo:lo + B // 2], half)
    for b in (0, B - 1):
        xb = xg[b:b + 1].permute(0, 3, 1, 2).cpu()
        _x3_check(big[b:b + 1], xb, w_fold, bias, None, 1, f"h2 tile, frame {b} of a 2.3 GB tensor")
    del xg, big, half
    # conv2: 64 -> 64, 3x3 stride 2 at 128x128 (pose_hrnet.py:282-284), 544 frames
    B, ci, co, hw = 544, 64, 64, 128
    x = torch.randn(8, ci, hw, hw, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    bn_cuda = tuple(t.cuda() for t in (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1,
                                       torch.rand(co, generator=g) * 0.4 + 0.8))
    wd, bd = capf.pack_conv(w.cuda(), bn_cuda)
    wp, bias = capf.pack_f32h2_gemm(w.cuda(), bn_cuda)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().repeat(B // 8, 1, 1, 1)
    xg += torch.arange(B, device="cuda").view(B, 1, 1, 1) * 1e-3
    assert xg.numel() * 4 > 2 ** 31
    big = capf.conv_nhwc_f32h2g(xg, wp, bias, 3, 2, 1, None, co)
    for lo in (0, B // 2):
        half = capf.conv_nhwc_f32h2g(xg[lo:lo + B // 2].contiguous(), wp, bias, 3, 2, 1, None, co)
        assert torch.equal(big[lo:lo + B // 2], half)
    w_fold = wd.cpu().double()[:, :9 * ci].view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
    for b in (0, B - 1):
        xb = xg[b:b + 1].permute(0, 3, 1, 2).cpu()
        _h2g_conv_check(big[b:b + 1], xb, w_fold, bd, None, 1, 2, f"h2g conv, frame {b} of a 2.3 GB tensor",
                        capf.conv_nhwc(xg[b:b + 1].contiguous(), wd, bd, 3, 2, 1, None))


# ---- round 6: a BasicBlock's conv1 -> conv2 tensor as split fp16 PLANES (igemm_f32h2_ws_tile.h, capf_op_conv_f32h2_planes) ----------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C", [(3, 64, 64, 32), (2, 32, 32, 64), (3, 16, 16, 128), (9, 8, 8, 256), (2, 24, 18, 64), (2, 64, 48, 32), (5, 12, 9, 128), (1, 96, 72, 48)])
@pytest.mark.parametrize("spread", [0, 8, 30])
def test_f32h2_planes_between_two_convs(B, H, W, C, spread):
    """conv1 (BN, ReLU) writes its output already split -- [piece 0 | piece 1] per pixel and 16-channel chunk, one power-of-two scale per
    (tile, chunk) -- and conv2 (BN, + residual, ReLU) stages the pieces as they are, moving the halo rows of the neighbouring tiles onto the
    chunk's scale.  Geometries: part-image tiles with neighbours above and below (64x64, 32x32, 64x48, 96x72), whole images, several images
    per tile, dealt-out pixel columns (widths off multiples of 16).  `spread`: the input's row bands are scaled by 2^+-spread, so that
    neighbouring tiles' outputs -- and their scales -- differ by up to 2^(2 spread): the halo rescale's normal, denormal (2^-15 .. 2^-24) and
    flush-to-zero branches all run.  Held to: the decoded planes == the fp32 route's conv1 output to 2^-22 relative (+ 2^-38 of the tile's
    largest value); conv2 on planes within 1e-6 of the sum of |terms| of an fp64 evaluation ON THE DECODED conv1 output (+ the block-scale
    term of include/capf.h's bound), and within 2e-6 of the all-fp32-tensor route."""
    from capf import lib as capf
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C + spread)
    x = torch.randn(B, C, H, W, generator=g) * torch.rand(B, C, H, W, generator=g).pow(2)
    if spread:
        band = torch.exp2(torch.randint(-spread, spread + 1, (B, 1, H, 1), generator=g).float())
        x = x * band
    convs = []
    for _ in range(2):
        w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
        bnp = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) * 0.4 + 0.8)
        bn_cuda = tuple(t.cuda() for t in bnp)
        wp, bias = capf.pack_conv_f32h2(w.cuda(), bn_cuda)
        wd, bd = capf.pack_conv(w.cuda(), bn_cuda)
        fold = wd.cpu().double()[:, :9 * C].view(C, 3, 3, C).permute(0, 3, 1, 2).contiguous()
        convs.append((wp, bias, fold))
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    # the fp32-tensor route: two launches of the same tile with plain tensors in between
    y1_f32, = capf.conv_nhwc_f32h2_group([(xg, convs[0][0], convs[0][1], 1, None, C)])
    y2_f32, = capf.conv_nhwc_f32h2_group([(y1_f32, convs[1][0], convs[1][1], 1, xg, C)])
    # the planes route
    y1_pl, exps = capf.conv_nhwc_f32h2_planes(xg, convs[0][0], convs[0][1], 1, None, C, planes_out=True)
    y2_pl = capf.conv_nhwc_f32h2_planes(y1_pl, convs[1][0], convs[1][1], 1, xg, C, exps_in=exps)
    assert torch.isfinite(y2_pl).all()
    tp = capf.f32h2_tile_pixels(B, H, W)
    dec = capf.planes_to_fp32(y1_pl, exps, tp)
    # (1) what the planes hold: conv1's fp32 result to 22 bits, tiny values to 2^-38 of their (tile, chunk)'s largest
    tile = (torch.arange(B * H * W, device="cuda") // tp)
    tmax = torch.zeros(int(tile.max()) + 1, C // 16, device="cuda")
    tmax.index_reduce_(0, tile, y1_f32.view(B * H * W, C // 16, 16).abs().amax(dim=2), "amax", include_self=True)
    tol = 2.0 ** -22 * y1_f32.abs() + 2.0 ** -38 * tmax[tile].repeat_interleave(16, dim=1).view(B, H, W, C)
    assert ((dec - y1_f32).abs() <= tol).all(), ((dec - y1_f32).abs() / tol).max().item()
    # (2) conv2 from the planes vs fp64 on the decoded operand
    dd = dec.cpu().double().permute(0, 3, 1, 2)
    fold, bias2 = convs[1][2], convs[1][1].double().cpu()
    want = F.relu(F.conv2d(dd, fold, bias2, 1, 1) + x.double())
    mass = F.conv2d(dd.abs(), fold.abs(), bias2.abs(), 1, 1) + x.double().abs()
    # block-scale term: M_c over the image (an over-estimate of the three tiles a tile's rows come from), W_c per 16-channel chunk
    m_c = dd.abs().view(B, C // 16, 16, H, W).amax(dim=(2, 3, 4))
    w_c = fold.abs().view(C, C // 16, 16 * 9).sum(dim=2)
    slack = 2.0 ** -38 * (m_c @ w_c.t())[:, :, None, None]
    got = y2_pl.double().cpu().permute(0, 3, 1, 2)
    err = (got - want).abs()
    assert (err <= 1e-6 * mass + slack).all(), (err / (1e-6 * mass + slack)).max().item()
    # (3) against the fp32-tensor route (whose conv2 reads conv1's fp32 values: one extra 2^-23 on every operand here)
    d_routes = ((y2_pl - y2_f32).double().cpu().permute(0, 3, 1, 2).abs() / (mass + slack * 2.0 ** 38 * 2.0 ** -23)).max().item()
    print(f"planes B={B} {H}x{W} C={C} spread 2^+-{spread}: conv2 vs fp64 on the decoded planes {(err / mass).max().item():.2e} of the sum of |terms|; vs the fp32-tensor route {d_routes:.2e}")
    assert d_routes <= 2e-6


@pytest.mark.parametrize("backbone,B,H,W", [("hrnet_32", 9, 256, 256), ("hrnet_32", 6, 192, 160), ("cpn", 5, 256, 192)])
def test_f32h2_unit_table_path_equals_the_computed_prologue(backbone, B, H, W):
    """Inside the engine the two-fp16-piece conv tile reads which bytes a lane stages from a per-geometry UNIT TABLE (igemm_f32h2_ws_tile.h, round 6);
    the stand-alone op (capf_op_conv_f32h2_group) has no table and computes the same addresses in its prologue.  Every such conv of a forward,
    recomputed with the stand-alone op from the operands the ENGINE produced, must give the engine's output bit for bit -- odd batches put a
    ragged last tile (which the table form leaves to the computed path) next to full ones; the maps cover one-segment-per-tile geometries
    (64^2 ... 16^2, 48 x 40 ...) and whole-images-per-tile ones (8^2, 6 x 5)."""
    from capf import lib as capf
    from capf import synth
    from test_gpu_fullsize import _model
    model, sd = _model(backbone, "fp32", 91)
    img, _, _ = synth.synth_inputs(B, H, W, seed=92, crop_range=(W, H))
    img_d = img.cuda()
    eng = model.engine_for(img_d)
    names = [n for n, _, _ in eng.schema()]
    table = eng.op_table(B)
    todo = [i for i, (_, k, _) in enumerate(table) if k == "igemm_f32h2_group_ws"]
    assert len(todo) >= (20 if B * H * W >= 9 * 256 * 256 else 4)      # (the tile takes a conv from 370 MFLOP)
    descs = {i: eng.op_describe(i) for i in todo}
    stream = torch.cuda.current_stream().cuda_stream
    checked, geoms = 0, set()
    for cp in sorted(set(descs[i].checkpoint for i in todo)):
        eng.forward_prefix(img_d, cp, stream)
        torch.cuda.synchronize()
        for i in [i for i in todo if descs[i].checkpoint == cp]:
            d = descs[i]
            x = eng.op_tensor(i, 0, (B, d.H, d.W, d.Cin), d.in_dtype).clone()
            res = eng.op_tensor(i, 4, (B, d.Ho, d.Wo, d.Cout), d.out_dtype).clone() if d.has_residual else None
            got = eng.op_tensor(i, 5, (B, d.Ho, d.Wo, d.Cout), d.out_dtype).clone()
            conv, bn = names[d.p_weight][:-len(".weight")], names[d.p_bn_weight][:-len(".weight")]
            wp, bias = capf.pack_conv_f32h2(sd[conv + ".weight"].cuda(), tuple(sd[bn + s].cuda() for s in (".weight", ".bias", ".running_mean", ".running_var")))
            want = capf.conv_nhwc_f32h2_group([(x, wp, bias, d.act, res, d.Cout)])[0]
            assert torch.equal(got, want), table[i][0]
            checked += 1
            geoms.add((d.H, d.W, d.Cin))
    print(f"{backbone} B={B} {H}x{W}: {checked} convs of the engine (unit tables) == the stand-alone op (computed prologue), bit for bit; geometries {sorted(geoms)}")
    assert len(geoms) >= (3 if B * H * W >= 9 * 256 * 256 else 1)
