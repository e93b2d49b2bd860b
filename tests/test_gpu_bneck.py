"""GPU: the first bottleneck of a layer1 under compute_dtype = bf16 as ONE kernel (csrc/bneck_bf16.hip; networks/resnet.py:58-93, 119-133;
pose_hrnet.py:98-136) -- conv1 -> conv2 -> conv3 + projection shortcut + ReLU with t1 / t2 / the shortcut on chip.

* which launches take it (capf_op_info), and that CAPF_PLAN_NO_BNECK / small batches keep the five launches;
* the product kernel's results are the TAP kernel's bit for bit -- the variant that also stores conv1's, conv2's and the shortcut's outputs
  and that the layer-wise tests (test_gpu_layerwise.py: every stage recomputed on the CPU from the operands the kernel itself produced) run;
* y against the five-launch plan on the same input: the same or the adjacent bf16 number almost everywhere (different accumulation order
  and bias placement; a flipped rounding of t1 / t2 moves a few outputs further), never a structured difference;
* ragged geometry: images whose last tiles touch the right / bottom border, and the zero padding of t1 (not of x) at the border."""
import pytest
import torch

from capf import synth
from capf.lib import PLAN_NO_BNECK
from test_gpu_fullsize import _model

pytestmark = pytest.mark.gpu

BLOCK = {"cpn": "backbone.resnet.layer1.0", "hrnet_48": "backbone.layer1.0"}
IDENT = {"cpn": ["backbone.resnet.layer1.1", "backbone.resnet.layer1.2"], "hrnet_48": ["backbone.layer1.1", "backbone.layer1.2", "backbone.layer1.3"]}


def _members(eng, backbone, B, block=None):
    t = eng.op_table(B)
    block = block or BLOCK[backbone]
    return {n[len(block) + 1:]: k for n, k, _ in t if n.startswith(block + ".")}


@pytest.mark.parametrize("backbone,B,H,W", [("cpn", 128, 384, 288), ("hrnet_48", 256, 256, 256)])
def test_first_bottleneck_is_one_launch_at_the_baseline_batches(backbone, B, H, W):
    model, _ = _model(backbone, "bf16", 5)
    img, k2d, kc = synth.synth_inputs(2, H, W, seed=6, crop_range=(W, H))
    eng = model.engine_for(img.cuda())
    m = _members(eng, backbone, B)
    assert set(m) == {"conv1", "conv2", "downsample.0", "conv3"} and set(m.values()) == {"bneck0_bf16<8x8>"}, m
    for blk in IDENT[backbone]:                   # the identity blocks behind it: conv1, conv2, conv3 + x as one launch each
        mi = _members(eng, backbone, B, blk)
        assert set(mi) == {"conv1", "conv2", "conv3"} and set(mi.values()) == {"bneck1_bf16<8x8>"}, mi
        assert "bneck1_bf16<8x8>" not in _members(eng, backbone, 2, blk).values()
    assert not any(k.startswith("igemm_bf16_pwchain") for _, k, _ in eng.op_table(B))      # (nothing left to chain: every layer1 block is one launch)
    assert "bneck0_bf16<8x8>" not in _members(eng, backbone, 2).values()          # (a handful of tiles per block: the five launches)
    off, _ = _model(backbone, "bf16", 5, PLAN_NO_BNECK)
    assert "bneck0_bf16<8x8>" not in _members(off.engine_for(img.cuda()), backbone, B).values()
    assert not any(k.startswith("bneck") for _, k, _ in off.engine_for(img.cuda()).op_table(B))
    fp, _ = _model(backbone, "fp32", 5)
    assert "bneck0_bf16<8x8>" not in _members(fp.engine_for(img.cuda()), backbone, B).values()


def _block_output(model, backbone, img_d, block=None):
    """y of the block after a prefix run that ends behind its conv3 (the TAP variant where the fused kernel runs), + the op's index."""
    eng = model.engine_for(img_d)
    B = img_d.shape[0]
    idx = [i for i, (n, _, _) in enumerate(eng.op_table(B)) if n == (block or BLOCK[backbone]) + ".conv3"][0]
    d = eng.op_describe(idx)
    eng.forward_prefix(img_d, idx + 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return eng.op_tensor(idx, 5, (B, d.Ho, d.Wo, d.Cout), d.out_dtype).clone(), idx, eng


@pytest.mark.parametrize("backbone,B,H,W", [("cpn", 24, 256, 192), ("cpn", 16, 384, 288), ("hrnet_48", 24, 256, 256), ("hrnet_48", 40, 192, 160)])
def test_fused_bottleneck_against_the_five_launches(backbone, B, H, W):
    """Same weights, same input, the block's output from both plans.  bf16 results of two correct evaluations differ where an fp32 pre-image
    lies next to a rounding boundary: one bf16 step almost everywhere that they differ at all, more only behind a flipped t1 / t2 element."""
    fused, _ = _model(backbone, "bf16", 11)
    plain, _ = _model(backbone, "bf16", 11, PLAN_NO_BNECK)
    img, _, _ = synth.synth_inputs(B, H, W, seed=12, crop_range=(W, H))
    img_d = img.cuda()
    _compare_block(fused, plain, backbone, img_d, None, "bneck0_bf16<8x8>", B, H, W)
    # ... and the LAST identity block: its input went through every fused block before it on one side, through the unfused plan on the other
    _compare_block(fused, plain, backbone, img_d, IDENT[backbone][-1], "bneck1_bf16<8x8>", B, H, W, chain=len(IDENT[backbone]) + 1)


def _compare_block(fused, plain, backbone, img_d, block, kernel, B, H, W, chain=1):
    y_f, idx, eng = _block_output(fused, backbone, img_d, block)
    assert eng.op_table(B)[idx][1] == kernel
    y_p, idx_p, eng_p = _block_output(plain, backbone, img_d, block)
    assert not eng_p.op_table(B)[idx_p][1].startswith("bneck")
    a, b = y_f.float(), y_p.float()
    diff = (a - b).abs()
    step = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-30          # one bf16 step at the value's magnitude (upper bound)
    frac_diff = (diff > 0).float().mean().item()
    frac_far = (diff > 2.0 * step).float().mean().item()
    rel = diff.max().item() / b.abs().max().item()
    print(f"{backbone} B={B} {H}x{W} {kernel} ({chain} fused blocks deep): {frac_diff:.3%} of the outputs differ, {frac_far:.4%} by more than two bf16 steps, "
          f"largest difference {rel:.2e} of the range")
    assert frac_diff <= 0.15 * chain and frac_far <= 2e-3 * chain and rel <= 2e-2 * chain
    # per-pixel structure: the border pixels (t1's zero padding) and the tile seams are not worse than the interior
    per_px = (diff > 2.0 * step).float().mean(dim=(0, 3))
    assert per_px.max().item() <= 0.02 * chain, per_px.max().item()


@pytest.mark.parametrize("backbone,B,H,W", [("cpn", 128, 384, 288), ("hrnet_48", 64, 256, 256)])
def test_product_kernel_equals_the_tap_kernel_bit_for_bit(backbone, B, H, W):
    """A full forward runs the product kernel (no taps), a debug forward and every prefix run the TAP variant: identical arithmetic, so the
    network's output and its four feature maps must be the same bits."""
    model, _ = _model(backbone, "bf16", 21)
    img, k2d, kc = synth.synth_inputs(B, H, W, seed=22, crop_range=(W, H))
    img_d, k2d_d = img.cuda(), k2d.cuda()
    eng = model.engine_for(img_d)
    with torch.no_grad():
        out_p = model(img_d, k2d_d, kc.clone().cuda()).clone()
        eng.set_debug(True)
        out_t = model(img_d, k2d_d, kc.clone().cuda()).clone()
        eng.set_debug(False)
    assert torch.equal(out_p, out_t)
    # and the block's y itself: prefix run (TAP) against a product launch that ends right behind the block is not expressible through the
    # ABI (a prefix is what selects TAP), so y is compared through the first consumer that both runs leave in the workspace: the output above
