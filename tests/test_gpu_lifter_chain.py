"""GPU: all res blocks of the lifter as ONE launch (csrc/lifter_chain.hip, OP_RES_CHAIN; pose_dformer.py:62-79, 231-234) against
  * the CPU oracle's token buffer after the res blocks (tokens_res), at batch 1 (three row tiles, a ragged last one), batch 7 and batch 64;
  * this library's own per-op route on the fp32 matrix pipe (CAPF_PLAN_NO_F32H2_GEMM: LayerNorm-folded qkv / fc1, attention, proj, fc2 as
    twenty launches) on the same frames -- the two routes share every expression per output, so they agree to fp32 summation order;
  * teacher-forced: the chain recomputed on the CPU from the ENGINE's own token buffer in front of it (tok_ctx), block by block.
The reference goldens (batch 2: tests/test_gpu_parity.py) run through the same launch and are unchanged."""
import contextlib
import copy
import io

import pytest
import torch

import capf_oracle as oracle
from capf import synth
from capf.lib import PLAN_NO_F32H2_GEMM

pytestmark = pytest.mark.gpu


def _run(batch, plan_flags=0, wseed=23, iseed=24):
    from mvn.models.conpose import CA_PF
    from mvn.utils.cfg import backbone_preset, config
    cfg = backbone_preset(copy.deepcopy(config), "hrnet_32")
    cfg.model.backbone.fix_weights = True
    with contextlib.redirect_stdout(io.StringIO()):
        model = CA_PF(cfg, plan_flags=plan_flags).eval()
    sd = synth.load_synthetic(model, seed=wseed, bn_mode="random")
    model = model.cuda()
    img, k2d, kc = synth.synth_inputs(batch, 256, 192, seed=iseed, crop_range=(192, 256))
    img_d = img.cuda()
    eng = model.engine_for(img_d)
    eng.set_debug(True)
    with torch.no_grad():
        out = model(img_d, k2d.cuda(), kc.clone().cuda())
    torch.cuda.synchronize()
    names = [n for n, _, _ in eng.op_table(batch)]
    return sd, (img, k2d, kc), eng, out.cpu(), names


@pytest.mark.parametrize("batch", [1, 7, 64])
def test_res_chain_matches_the_oracle_and_the_per_op_route(batch):
    sd, (img, k2d, kc), eng, out, names = _run(batch)
    assert names.count("res.chain") == 1 and not any(n.startswith("res0.") for n in names)
    assert [n for n in names if n.endswith(".mlp")] == [f"ctx{i}.mlp" for i in range(4)] and "ctx0.fc1" not in names
    tok_ctx = eng.tensor("tok_ctx").cpu()                                    # [B, 17, 5, 128]: what the chain read
    tok_res = eng.tensor("tok_res").cpu()
    sd2, _, eng2, out2, names2 = _run(batch, PLAN_NO_F32H2_GEMM)
    assert "res.chain" not in names2 and names2.count("res0.qkv") == 1
    per_op = eng2.tensor("tok_res").cpu()
    scale = per_op.abs().max().item()
    d_routes = (tok_res - per_op).abs().max().item()
    d_ctx = (tok_ctx - eng2.tensor("tok_ctx").cpu()).abs().max().item()      # the context blocks: ctx_attn + ONE launch for the MLP half vs + two
    # teacher-forced: the oracle's res blocks on the engine's own input rows
    with torch.no_grad():
        x = tok_ctx.reshape(batch * 17, 5, 128).clone()
        for i in range(4):
            x = oracle._attn_block(sd, f"volume_net.res_blocks.{i}", x, 8)
    d_forced = (tok_res.reshape(batch * 17, 5, 128) - x).abs().max().item()
    taps = {}
    frames = list(range(min(batch, 4)))
    with torch.no_grad():
        want = oracle.ca_pf_forward(sd, img[frames], k2d[frames], kc[frames].clone(), backbone="hrnet_32", taps=taps)
    d_oracle = (tok_res[frames].reshape(len(frames), 17, -1) - taps["tokens_res"]).abs().max().item()
    d_out = (out[frames] - want).abs().max().item()
    print(f"batch {batch}: token buffer behind the context blocks, fused MLP halves vs the fp32 pipe {d_ctx:.2e}; behind the res blocks (max |x| {scale:.2f}) -- one launch vs twenty on the fp32 pipe {d_routes:.2e}; vs the oracle's "
          f"blocks on the engine's own rows {d_forced:.2e}; vs the oracle end to end {d_oracle:.2e}; joints {d_out:.2e} m")
    assert d_ctx <= 2e-5 * scale and d_routes <= 2e-5 * scale and d_forced <= 2e-5 * scale and d_oracle <= 1e-4 * scale and d_out <= 1e-3
