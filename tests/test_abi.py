"""CPU: the C-ABI library loads, exports every symbol include/capf.h declares, and its parameter
schema / host module state_dict equal the reference's state_dict (captured in tests/golden/schema_*.json
by oracle/make_goldens.py).  No compute calls — there is no GPU here."""
import copy
import json
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "capf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(capf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import capf
    from capf.lib import EXPORTS
    lib = capf.load_library()
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(lib, sym), f"libcapf.so does not export {sym}"
    assert sorted(EXPORTS) == declared
    assert b"gfx950" in lib.capf_version()


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + csrc/capf.map: the dynamic symbol table is the header, not the C++ internals (`nm -D` showed
    capf::launch_* beside the 81 C symbols)."""
    import shutil
    import subprocess
    from capf import lib as capf_lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", capf_lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == _header_symbols()


def _cfg(backbone):
    from mvn.utils.cfg import backbone_preset, config
    c = backbone_preset(copy.deepcopy(config), backbone)
    c.model.backbone.fix_weights = True
    return c


@pytest.mark.parametrize("backbone", ["hrnet_32", "hrnet_48", "cpn"])
def test_schema_equals_reference_state_dict(backbone):
    from capf import Engine
    from mvn.models import _native
    want = json.load(open(os.path.join(ROOT, "tests", "golden", f"schema_{backbone}.json")))
    c = _native.make_capf_config(_cfg(backbone), 256, 192)
    eng = Engine(c, device=None)
    got = {n: list(s) for n, s, k in eng.schema()}
    assert got == want
    launches, flops = eng.stats(1)
    assert launches > 100 and flops > 1e10
    assert eng.workspace_bytes(4) > eng.workspace_bytes(1) > 0          # inference + training regions
    layout, total = eng.grad_layout()
    lifter = {n: s for n, s in want.items() if n.startswith("volume_net.")}
    assert set(layout) == set(lifter) and total == sum(int(__import__("math").prod(s)) for s in lifter.values())
    c.training = 0
    inf = Engine(c, device=None)
    assert inf.workspace_bytes(4) == 4 * inf.workspace_bytes(1) > 0      # activations scale with the batch


@pytest.mark.parametrize("backbone", ["hrnet_32", "cpn"])
def test_host_module_state_dict_and_freeze(backbone):
    import contextlib, io
    from mvn.models.conpose import CA_PF
    want = json.load(open(os.path.join(ROOT, "tests", "golden", f"schema_{backbone}.json")))
    with contextlib.redirect_stdout(io.StringIO()):
        m = CA_PF(_cfg(backbone))
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == want
    assert not any(p.requires_grad for p in m.backbone.parameters())          # conpose.py:22-25
    assert all(p.requires_grad for p in m.volume_net.parameters())
    m.load_state_dict(sd, strict=True)
    # callers' access patterns (train.py:147-148, 339): .backbone.eval(), .volume_net.named_parameters()
    m.backbone.eval(); m.volume_net.train()
    assert len(list(m.volume_net.named_parameters())) == 191 - 0 if backbone == "hrnet_32" else True


def test_flops_match_survey():
    """Algorithmic FLOPs reported by the plan == SURVEY.md §8d (probed on the reference)."""
    from capf import Engine
    from mvn.models import _native
    for bb, hw, gflop in (("hrnet_32", (256, 256), 20.995), ("hrnet_48", (256, 256), 42.469), ("cpn", (384, 288), 23.182)):
        eng = Engine(_native.make_capf_config(_cfg(bb), *hw), device=None)
        _, f = eng.stats(1)
        assert abs(f / 1e9 - gflop) / gflop < 0.01, (bb, f / 1e9)


def test_errors_are_reported():
    from capf import CapfError, Engine
    from mvn.models import _native
    c = _native.make_capf_config(_cfg("hrnet_32"), 250, 192)      # not a multiple of 32
    with pytest.raises(CapfError):
        Engine(c, device=None)
    bad = _cfg("hrnet_32")
    bad.model.backbone.STAGE3.NUM_BLOCKS = [4, 4]
    with pytest.raises(ValueError):                                # pose_hrnet.py:159-175 semantics
        _native.make_capf_config(bad)
    import torch
    from mvn.models.conpose import CA_PF
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = CA_PF(_cfg("hrnet_32"))
    with pytest.raises(CapfError):                                 # no CPU fallback
        m(torch.zeros(1, 256, 192, 3), torch.zeros(1, 17, 2), torch.zeros(1, 17, 2))


@pytest.mark.parametrize("backbone", ["hrnet_32", "hrnet_48", "cpn"])
def test_region_schedule_is_a_valid_topological_order(backbone):
    """capf_set_lanes mode 2 issues every fork/join region level by level (the fp32 convs of a level in one
    grouped launch).  Host-side check on the static plan: inside a region an op's level is above the level of
    every earlier op it conflicts with on a workspace buffer, and the ops of one level are pairwise
    independent (so one grid may run them concurrently)."""
    from capf import Engine
    from mvn.models import _native
    eng = Engine(_native.make_capf_config(_cfg(backbone), 256, 192), device=None)
    sched = eng.op_schedule()
    regions = {}
    for i, (rg, lv, ln, rd, wr) in enumerate(sched):
        if rg >= 0:
            assert lv >= 0, f"op {i} of region {rg} has no level"
            regions.setdefault(rg, []).append(i)
    assert len(regions) >= (20 if backbone != "cpn" else 5)
    widest = 0
    for rg, ops in regions.items():
        by_level = {}
        for a_pos, a in enumerate(ops):
            _, la, _, ra, wa = sched[a]
            by_level.setdefault(la, []).append(a)
            for b in ops[:a_pos]:                       # b precedes a in program order
                _, lb, _, rb, wb = sched[b]
                conflict = (set(wb) & (set(ra) | set(wa))) or (set(rb) & set(wa))
                if conflict:
                    assert la > lb, f"op {a} (level {la}) depends on op {b} (level {lb}) via buffers {conflict}"
        for lv, members in by_level.items():
            widest = max(widest, len(members))
            for x in members:
                for y in members:
                    if x < y:
                        assert not (set(sched[x][4]) & (set(sched[y][3]) | set(sched[y][4])))
                        assert not (set(sched[y][4]) & set(sched[x][3]))
    assert widest >= 4          # e.g. the four branches of a stage-4 module / the CPN refine cascades


def test_tile_choice_avoids_a_second_round_for_the_lifter_gemms():
    """Static plan, no GPU: the cost model behind pick_tile (igemm_f32.hip) must not give the lifter's
    M = 17 * 64 = 1088-row GEMMs 128-row tiles (9 x 30 = 270 tiles = two rounds on 256 CUs); the 32-channel
    64x64 branch keeps its tall 256x32 tile when it is launched on its own."""
    from capf import Engine
    from mvn.models import _native
    from capf.lib import PLAN_NO_F32H2_GEMM
    eng = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256), device=None)
    table = {name: kern for name, kern, _ in eng.op_table(64)}
    # round 5: from batch 5 the lifter's projections run the two-fp16-piece GEMM (64 x 64 tiles as well), the LayerNorm-folded ones
    # (K = 128: res blocks' qkv / fc1, context blocks' fc1) included; every batch below 5 stays on the fp32 MFMA kernel
    assert table["joint0.qkv"] == "igemm_f32h2g<64x64,rows>" and table["joint0.fc2"] == "igemm_f32h2g<64x64,rows>"
    assert table["ctx0.mlp"] == "mlp_chain" and "ctx0.fc1" not in table       # (round 6: a context block's MLP half is one launch)
    # round 6: the res blocks are ONE launch at every batch (lifter_chain.hip) ...
    assert table["res.chain"] == "res_chain" and not any(n.startswith("res0.") for n in table)
    assert {name: kern for name, kern, _ in eng.op_table(1)}["res.chain"] == "res_chain"
    assert {name: kern for name, kern, _ in eng.op_table(4)}["joint0.qkv"].startswith("igemm_f32<")
    eng_f = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_NO_F32H2_GEMM), device=None)
    table_f = {name: kern for name, kern, _ in eng_f.op_table(64)}
    assert table_f["joint0.qkv"] == "igemm_f32<w4,64x64,rows>"
    assert table_f["res0.qkv"].startswith("igemm_f32<") and "res.chain" not in table_f       # ... unless the plan has no two-piece packs
    assert table_f["ctx0.fc1"].startswith("igemm_f32<") and "ctx0.mlp" not in table_f
    assert table_f["joint0.fc2"] == "igemm_f32<w4,64x64,rows>"
    assert not any(k.startswith("igemm_f32h2g") for k in table_f.values())
    # the 3x3 stride-1 convs run the direct kernel (tall 256x32 tile for the 32-channel 64x64 branch launched on its own) below batch 5 and
    # the split-fp32 tile from 370 MFLOP per conv (batch 5 for these branches); a plan without that tile (CAPF_PLAN_NO_F32X3) runs them on
    # the Winograd kernel from batch 24
    assert table["backbone.stage2.0.branches.0.0.conv1"] == "igemm_f32h2_group_ws"       # (round 5: two fp16 pieces, three products)
    assert {name: kern for name, kern, _ in eng.op_table(6)}["backbone.stage2.0.branches.0.0.conv1"] == "igemm_f32h2_group_ws"
    from capf.lib import PLAN_NO_F32X3, PLAN_F32X3_EXACT
    eng_x = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_F32X3_EXACT), device=None)   # round 4's exact three-piece tile
    assert {name: kern for name, kern, _ in eng_x.op_table(64)}["backbone.stage2.0.branches.0.0.conv1"] == "igemm_f32x3_group_ws"
    assert [k for _, k, _ in eng_x.op_table(64) if not k.startswith("igemm_f32x3")] == [k for _, k, _ in eng.op_table(64) if not k.startswith("igemm_f32h2_")]
    eng_w = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_NO_F32X3), device=None)
    assert {name: kern for name, kern, _ in eng_w.op_table(24)}["backbone.stage2.0.branches.0.0.conv1"] == "igemm_wino43_group"   # F(4,3): row length 64 is a multiple of 4
    # (between batch 5 and the Winograd kernels' batch 24 the direct route is the two-fp16-piece GEMM; the whole fp32-pipe plan of round 3
    # is CAPF_PLAN_NO_F32X3 | CAPF_PLAN_NO_F32H2_GEMM)
    assert {name: kern for name, kern, _ in eng_w.op_table(16)}["backbone.stage2.0.branches.0.0.conv1"].startswith("igemm_f32h2g<")
    eng_p = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM), device=None)
    assert {name: kern for name, kern, _ in eng_p.op_table(16)}["backbone.stage2.0.branches.0.0.conv1"].startswith("igemm_f32<")
    assert not any(k.startswith(("igemm_f32h2", "igemm_f32x3")) for _, k, _ in eng_p.op_table(64))
    small = {name: kern for name, kern, _ in eng.op_table(4)}
    assert small["backbone.stage2.0.branches.0.0.conv1"].startswith("igemm_f32<")
    # layer1's HBM-bound 1x1 bottleneck convs: the pointwise kernel from 2048 tiles per launch, the general tile below
    assert table["backbone.layer1.0.conv3"] == "igemm_f32_pw<w4,128x64>"
    assert table["backbone.layer1.1.conv1"] == "igemm_f32_pw<w4,128x64>"
    # (the conv3 -> next conv1 pairs stay on the fp32 kernels at every batch: the chained and the two-launch routes must give the same bits)
    assert {name: kern for name, kern, _ in eng.op_table(8)}["backbone.layer1.0.conv3"].startswith("igemm_f32<")
    assert {name: kern for name, kern, _ in eng.op_table(8)}["backbone.layer1.0.conv1"].startswith("igemm_f32h2g<")
    assert {name: kern for name, kern, _ in eng_f.op_table(8)}["backbone.layer1.0.conv1"].startswith("igemm_f32<")
    big = {name: kern for name, kern, _ in eng.op_table(512)}
    assert big["joint0.qkv"] == "igemm_f32h2g<128x128,rows>" and big["joint0.fc1"] == "igemm_f32h2g<128x64,rows>"     # (1020 / 680 tiles of 128 x 128)
    assert {name: kern for name, kern, _ in eng_f.op_table(512)}["joint0.qkv"] in ("igemm_f32<w4,128x128,rows>", "igemm_f32<w4,128x64,rows>")


def test_conv_group_rejects_bad_arguments_without_a_gpu():
    """capf_op_conv_group validates before it launches: the error paths need no device."""
    import ctypes
    import capf
    from capf.lib import ConvDesc
    lib = capf.load_library()
    lib.capf_op_conv_group.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ConvDesc)]
    lib.capf_op_conv_group.restype = ctypes.c_int
    d = (ConvDesc * 9)()
    for i in range(9):
        d[i].x = d[i].w_packed = d[i].bias = d[i].y = 4096          # never dereferenced on these paths
        d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].Cout, d[i].ks, d[i].stride, d[i].act = 1, 8, 8, 32, 32, 3, 1, 0
    assert lib.capf_op_conv_group(None, 0, d) != 0                  # empty group
    assert lib.capf_op_conv_group(None, 9, d) != 0                  # more than MAXG = 8 problems
    assert lib.capf_op_conv_group(None, 2, None) != 0
    d[1].Cin = 3                                                    # the small-Cin stem kernel cannot be grouped
    assert lib.capf_op_conv_group(None, 2, d) != 0
    d[1].Cin, d[1].ks = 32, 7                                       # 49 taps do not fit the 32-bit tap mask
    assert lib.capf_op_conv_group(None, 2, d) != 0


def test_integration_stub_struct_matches_the_header_and_the_binding():
    """INTEGRATION.md's ctypes stub, include/capf.h's capf_config and capf.lib.CapfConfig must list the same fields in the
    same order (round 1 shipped a stub one field short: the library would have read past the caller's struct)."""
    import ctypes, os, re
    from capf.lib import CapfConfig
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bound = [n for n, _ in CapfConfig._fields_]
    header = open(os.path.join(root, "include", "capf.h")).read()
    body = header[header.index("typedef struct capf_config {"):header.index("} capf_config;")]
    decl = []
    for line in body.splitlines()[1:]:
        m = re.match(r"\s*int32_t\s+([^;]+);", line)
        if m:
            decl += [re.sub(r"\[\d+\]", "", v).strip() for v in m.group(1).split(",")]
    assert decl == bound
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = doc[doc.index("class capf_config(ctypes.Structure)"):doc.index("class CA_PF(nn.Module):")]
    assert re.findall(r'\("(\w+)", ctypes\.c_int32', stub) == bound
    assert ctypes.sizeof(CapfConfig) == 4 * 24
    ctor = re.search(r"c = capf_config\(([^\n]*)\)\s+#", doc).group(1)
    assert len(re.sub(r"\([^)]*\)", "T", ctor).split(",")) == len(bound)


def test_hot_kernels_do_not_spill_registers():
    """Build guard: a spilling MFMA kernel still produces right answers, 5-8x slower (scratch traffic and slow dispatch) --
    the grouped bf16 kernel once went from 16600 to 2000 frames/s that way.  Compile the conv / GEMM sources to assembly for
    gfx950 and require vgpr_spill_count == 0 for every kernel."""
    import concurrent.futures, re, shutil, subprocess, tempfile
    hipcc = shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("hipcc not on PATH")
    csrc = os.path.join(ROOT, "contextaware-poseformer_amd", "csrc")

    def spills(name):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, name + ".s")
            subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                            "-S", "--cuda-device-only", os.path.join(csrc, name + ".hip"), "-o", out],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            text = open(out).read()
        names = re.findall(r"^\s+\.name:\s+(\S+)", text, flags=re.M)
        counts = [int(v) for v in re.findall(r"^\s+\.vgpr_spill_count:\s+(\d+)", text, flags=re.M)]
        assert len(names) == len(counts) and counts
        return [(n, c) for n, c in zip(names, counts) if c]

    with concurrent.futures.ThreadPoolExecutor(4) as ex:
        bad = sum(ex.map(spills, ["igemm_bf16", "igemm_f32", "igemm_f32_pw", "igemm_f32_pwchain", "igemm_bf16_pwchain", "igemm_wino"]), [])
    assert not bad, f"kernels with register spills: {bad}"


def test_plan_flags_select_kernel_families_and_nothing_else_does(monkeypatch):
    """capf_config::plan_flags is the ONLY way to take a kernel family out of the plan: the A/B environment switches of
    earlier rounds are compiled out of the product library (kernels.h diag_env), unknown flag bits are rejected."""
    from capf import Engine
    from capf.lib import PLAN_F32X3_EXACT, PLAN_NO_F32H2_GEMM, PLAN_NO_F32X3, PLAN_NO_FUSED_LIFTER, PLAN_NO_WINOGRAD, CapfError
    from mvn.models import _native

    def kernels(flags, embed=128):
        cfg = _cfg("hrnet_32")
        cfg.model.poseformer.embed_dim_ratio = embed
        eng = Engine(_native.make_capf_config(cfg, 256, 256, plan_flags=flags), device=None)
        return [(n, k) for n, k, _ in eng.op_table(64)]

    base = kernels(0)
    assert any(k == "ctx_attn" for _, k in base) and any(k.startswith("igemm_f32h2_") for _, k in base) and any(k.startswith("igemm_f32h2g") for _, k in base)
    assert not any(k.startswith("igemm_f32x3") for _, k in base)
    exact = kernels(PLAN_F32X3_EXACT)                                       # round 4's tile for the 3x3 convs: the exact three-bf16-piece split
    assert [n for n, k in exact if k.startswith("igemm_f32x3")] == [n for n, k in base if k.startswith("igemm_f32h2_")]
    no_x3 = kernels(PLAN_NO_F32X3)                                          # 3x3 convs on the Winograd kernels at every batch
    assert any(k.startswith("igemm_wino") for _, k in no_x3) and not any(k.startswith(("igemm_f32x3", "igemm_f32h2_")) for _, k in no_x3)
    no_g = kernels(PLAN_NO_F32H2_GEMM)                                      # the other convs / linears on the fp32 matrix pipe
    assert not any(k.startswith("igemm_f32h2g") for _, k in no_g) and [n for n, k in no_g if k.startswith("igemm_f32h2_")] == [n for n, k in base if k.startswith("igemm_f32h2_")]
    r3 = kernels(PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM)                        # round 3's plan: everything on the fp32 pipe
    assert not any(k.startswith(("igemm_f32x3", "igemm_f32h2")) for _, k in r3)
    for var in ("CAPF_LIFTER_FUSED", "CAPF_WINO", "CAPF_BF16_RH", "CAPF_WINO_F43", "CAPF_F32X3_MIN_MFLOP"):
        monkeypatch.setenv(var, "0")
    assert kernels(0) == base                                               # the environment does not reach the product plan
    unfused = kernels(PLAN_NO_FUSED_LIFTER)
    assert not any(k in ("ctx_attn", "embed") for _, k in unfused) and any(k == "deform_sample" for _, k in unfused)
    assert not any(k.startswith(("igemm_wino", "igemm_f32x3", "igemm_f32h2_")) for _, k in kernels(PLAN_NO_WINOGRAD))
    with pytest.raises(CapfError):
        kernels(1 << 14)
    # embed_dim_ratio beyond the fused kernels' register / LDS budget: the plan falls back to one kernel per op
    wide = kernels(0, embed=288)
    assert not any(k in ("ctx_attn", "embed") for _, k in wide) and any(k == "deform_sample" for _, k in wide)


def test_first_bottleneck_plan_bf16_without_a_gpu():
    """bneck_bf16.hip in the plan (no device needed): layer1.0's four convs name one kernel at the baseline batches of the two bf16 configurations,
    CAPF_PLAN_NO_BNECK / fp32 / small batches keep the five launches, and the five tensors the kernel touches never share workspace."""
    from capf import Engine
    from capf.lib import PLAN_NO_BNECK
    from mvn.models import _native
    for backbone, H, W, B, blk in (("cpn", 384, 288, 128, "backbone.resnet.layer1.0"), ("hrnet_48", 256, 256, 256, "backbone.layer1.0")):
        eng = Engine(_native.make_capf_config(_cfg(backbone), H, W, compute_dtype="bf16"), device=None)
        mem = {n: k for n, k, _ in eng.op_table(B) if n.startswith(blk + ".")}
        assert len(mem) == 4 and set(mem.values()) == {"bneck0_bf16<8x8>"}, mem
        ident = {n: k for n, k, _ in eng.op_table(B) if ".layer1." in n and not n.startswith(blk + ".")}
        assert len(ident) in (6, 9) and set(ident.values()) == {"bneck1_bf16<8x8>"}, ident       # (the identity blocks: conv1, conv2, conv3 + x per launch)
        assert not any(k == "bneck0_bf16<8x8>" for _, k, _ in eng.op_table(2))
        off = Engine(_native.make_capf_config(_cfg(backbone), H, W, compute_dtype="bf16", plan_flags=PLAN_NO_BNECK), device=None)
        assert not any(k.startswith("bneck") for _, k, _ in off.op_table(B))
        f32 = Engine(_native.make_capf_config(_cfg(backbone), H, W), device=None)
        assert not any(k.startswith("bneck") for _, k, _ in f32.op_table(B))
        # the block's conv1 / conv2 / shortcut are checkpointed behind its conv3 (the layer-wise tests read them from the TAP kernel's stores)
        names = [n for n, _, _ in eng.op_table(B)]
        c3 = names.index(blk + ".conv3")
        for m in ("conv1", "conv2", "downsample.0"):
            assert eng.op_describe(names.index(f"{blk}.{m}")).checkpoint == c3 + 1


def test_executed_flops_of_winograd_ops_are_half_or_two_thirds_of_the_algorithmic_count():
    from capf import Engine
    from mvn.models import _native
    eng = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256), device=None)
    # batch 64: the branch convs run the split-fp32 tile -- three fp16 piece products per fp32 product (six bf16 ones under
    # CAPF_PLAN_F32X3_EXACT), counted on the 16-bit pipe
    t64, e64 = eng.op_table(64), eng.op_executed_flops(64)
    assert any(k.startswith("igemm_f32h2_") for _, k, _ in t64) and any(k.startswith("igemm_f32h2g") for _, k, _ in t64)
    assert all(abs(e / a - 3.0) < 1e-9 for (n, k, a), e in zip(t64, e64) if k.startswith("igemm_f32h2_"))
    assert all(3.0 - 1e-9 <= e / a <= 3.6 for (n, k, a), e in zip(t64, e64) if k.startswith("igemm_f32h2g"))        # (K padded to 32)
    from capf.lib import PLAN_NO_F32H2_GEMM, PLAN_NO_F32X3, PLAN_F32X3_EXACT
    eng_x = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_F32X3_EXACT), device=None)
    t64, e64 = eng_x.op_table(64), eng_x.op_executed_flops(64)
    assert any(k.startswith("igemm_f32x3") for _, k, _ in t64)
    assert all(abs(e / a - 6.0) < 1e-9 for (n, k, a), e in zip(t64, e64) if k.startswith("igemm_f32x3"))
    eng_w = Engine(_native.make_capf_config(_cfg("hrnet_32"), 256, 256, plan_flags=PLAN_NO_F32X3 | PLAN_NO_F32H2_GEMM), device=None)
    table, ex = eng_w.op_table(32), eng_w.op_executed_flops(32)
    seen = set()
    for (name, kern, alg), e in zip(table, ex):
        if kern.startswith("igemm_wino"):
            r = e / alg
            assert abs(r - 0.5) < 1e-9 or abs(r - 2.0 / 3.0) < 1e-9, (name, r)
            seen.add(round(r, 3))
        elif kern.startswith("igemm") and alg > 0:
            assert 1.0 - 1e-9 <= e / alg <= 1.2, (name, kern, e / alg)       # K padded to the chunk width only
    assert 0.5 in seen
    # below the Winograd batch threshold the same ops run the direct kernel: executed == algorithmic
    t1, e1 = eng.op_table(1), eng.op_executed_flops(1)
    assert all(abs(e / a - 1.0) < 0.2 for (n, k, a), e in zip(t1, e1) if k.startswith("igemm") and a > 0)


def test_sync_batchnorm_conversion_keeps_the_reference_state_dict():
    """train.py:317-318 on the host mirror (CPU part; the GPU box checks that the engine binds the converted tree)."""
    import contextlib, io
    import torch
    from mvn.models.conpose import CA_PF
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "schema_hrnet_32.json")))
    with contextlib.redirect_stdout(io.StringIO()):
        m = CA_PF(_cfg("hrnet_32"))
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == want
    assert sum(isinstance(x, torch.nn.SyncBatchNorm) for x in m.modules()) == 292
    assert not any(p.requires_grad for p in m.backbone.parameters())


def test_depth_is_a_plan_parameter_of_the_variant_without_context_blocks():
    """capf_config.depth (ContextPose_mpi/model/pose_dformer.py:199): blocks per group of the MPI-INF-3DHP lifter.  0 = levels;
    the H36M model and the training path are built for depth == levels only and say so."""
    from capf import Engine
    from capf.lib import CapfError
    from mvn.models import _native

    def plan(depth, context_blocks, training=0):
        c = _native.make_capf_config(_cfg("hrnet_32"), 256, 192, context_blocks=context_blocks)
        c.depth, c.training = depth, training
        eng = Engine(c, device=None)
        names = [n for n, _, _ in eng.op_table(2)]
        chain_flops.append(sum(f for n, _, f in eng.op_table(2) if n == "res.chain"))
        schema = [s[0] for s in eng.schema()]
        eng.close()
        return names, schema

    chain_flops = []
    for depth, want in ((0, 4), (4, 4), (2, 2), (6, 6)):
        names, schema = plan(depth, False)
        # (round 6: the res blocks of any depth are ONE launch, lifter_chain.hip -- its FLOPs count the blocks)
        assert names.count("res.chain") == 1 and chain_flops[-1] > 0 and abs(chain_flops[-1] / want - chain_flops[0] / 4) < 1e-6 * chain_flops[0]
        assert sum(n.endswith(".qkv") and n.startswith("joint") for n in names) == want
        assert sum(s.endswith("attn.qkv.weight") for s in schema) == 2 * want
    for bad in ((2, True, 0), (2, False, 1), (9, False, 0), (-1, False, 0)):
        with pytest.raises(CapfError):
            plan(*bad)
