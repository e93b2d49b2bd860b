"""Shared by the bf16 GPU tests: per-stage attribution of a compute_dtype='bf16' run and the bounds it is held to."""

# Parity bounds of the bf16 mode, against the bf16-EMULATING oracle (oracle.ca_pf_forward(..., emulate_bf16=True)): the HIP
# path and the emulation round the same values at the same places, so they differ only through fp32 summation order and the
# occasional bf16 rounding flip that order causes (one flip = 2^-9 relative on one activation).  The joints get the
# north-star tolerance of the fp32 path (1e-3 m); maps 2e-3 relative L2 (a quarter of one bf16 ulp).  The distance to the
# fp32 oracle is the mode's ROUNDING BUDGET: printed per stage and only sanity-capped (3e-2 m: 8 mantissa bits, ~300 layers).
BF16_EMU_JOINTS = 1e-3
BF16_EMU_MAPS = 2e-3
BF16_BUDGET_CAP = 3e-2


def bf16_stage_report(tag, eng, got, rows, taps_emu, want_emu, taps_f32, want_f32):
    """eng: engine after a debug forward of the FULL batch; `rows`: indices of the frames the oracles ran (list) or None = all.
    Every tap against (a) the bf16-emulating oracle (the parity bound) and (b) the fp32 oracle (the rounding budget).
    Returns {stage: (err vs emulation, err vs fp32)}; maps are relative L2, tokens / joints max-abs."""
    import torch
    sel = (lambda t: t) if rows is None else (lambda t: t[rows])
    n = got.shape[0] if rows is None else len(rows)
    rep = {}
    for l in range(4):
        f = sel(eng.tensor(f"feat{l}").float().cpu()).permute(0, 3, 1, 2)
        rep[f"feat{l}"] = tuple(((f - t["features"][l]).norm() / t["features"][l].norm()).item() for t in (taps_emu, taps_f32))
    if "tokens_ctx" in taps_emu:
        tk = sel(eng.tensor("tok_ctx").cpu()).permute(0, 2, 1, 3)
        rep["tok_ctx"] = tuple((tk - t["tokens_ctx"]).abs().max().item() for t in (taps_emu, taps_f32))
    tk = sel(eng.tensor("tok_res").cpu()).reshape(n, 17, -1)
    rep["tok_res"] = tuple((tk - t["tokens_res"]).abs().max().item() for t in (taps_emu, taps_f32))
    tk = sel(eng.tensor("tok_joint").cpu()).reshape(n, 17, -1)
    rep["tok_joint"] = tuple((tk - t["tokens_joint"]).abs().max().item() for t in (taps_emu, taps_f32))
    g = sel(got)
    rep["joints"] = ((g - want_emu).abs().max().item(), (g - want_f32).abs().max().item())
    rep["joints_mean_dist"] = ((g - want_emu).norm(dim=-1).mean().item(), (g - want_f32).norm(dim=-1).mean().item())
    print(f"{tag}: stage          vs bf16-emulating oracle   vs fp32 oracle (rounding budget)")
    for k, (a, b) in rep.items():
        print(f"    {k:18s} {a:12.3e} {b:24.3e}")
    return rep


def check_bf16_report(rep):
    for l in range(4):
        assert rep[f"feat{l}"][0] <= BF16_EMU_MAPS, (l, rep[f"feat{l}"])
    assert rep["joints"][0] <= BF16_EMU_JOINTS, rep["joints"]
    assert rep["joints"][1] <= BF16_BUDGET_CAP, rep["joints"]
