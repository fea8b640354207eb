"""Model-level parity on a real MI355X: the drop-in modules (Transformer3DModel.forward,
BasicTransformerBlock, CausalVideoAutoencoder.decode, the denoise loop) against the CPU oracle
on the same seeded inputs and weights.

Tolerance, per BASELINE.json (rtol = 2e-3 in bf16 per denoise step against the reference's
eager path): both our result and the reference's bf16 eager result are bf16 renderings of the
same fp32 computation, so the test measures each against the fp32 oracle ("truth") and requires

        err(ours) <= err(reference bf16 eager) + 2e-3          (relative L2 over the tensor)

where the reference's bf16 eager path is the oracle run in bf16 -- it has the reference's
rounding points and is pinned to the reference's own bf16 output by
tests/test_oracle_golden.py::test_g5_transformer_bf16_twin.
"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16
RTOL = 2e-3


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def maxrel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def assert_parity(out, truth, eager, what):
    """err(ours) <= err(reference bf16 eager) + RTOL in relative L2, plus an element-wise guard: our largest error
    (as a fraction of the output range) may not exceed the eager path's largest error by more than half of it + 1 %."""
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite output"
    e_ours, e_ref = rel(out, truth), rel(eager, truth)
    m_ours, m_ref = maxrel(out, truth), maxrel(eager, truth)
    e_direct = rel(out, eager)            # ours against the bf16 eager rendering itself (two bf16 renderings: not asserted)
    print(f"{what}: rel L2 vs fp32 oracle: ours {e_ours:.3e} / reference-bf16-eager {e_ref:.3e};  ours vs bf16-eager directly "
          f"{e_direct:.3e};  max err / range ours {m_ours:.3e} / eager {m_ref:.3e}")
    assert e_ours <= e_ref + RTOL, (what, e_ours, e_ref)
    assert m_ours <= 1.5 * m_ref + 1e-2, (what, m_ours, m_ref)
    return e_ours, e_ref


def dit_case(heads, dh, layers, grid, B, T, caption=128, seed=0, per_token=False):
    from oracle import dit, sched
    cfg = dict(dit.default_2b_config(), num_attention_heads=heads, attention_head_dim=dh, num_layers=layers,
               cross_attention_dim=heads * dh, caption_channels=caption)
    sd32 = {k: v.to(BF).float() for k, v in dit.init_state_dict(cfg, seed=seed).items()}   # bf16-representable
    f, h, w = grid
    N = f * h * w
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.randn(B, N, cfg["in_channels"], generator=g).to(BF)
    enc = torch.randn(B, T, caption, generator=g).to(BF)
    mask = torch.ones(B, T)
    mask[:, T - T // 3:] = 0
    if per_token:
        ts = torch.full((B, N), 0.7)
        ts[:, : h * w] = 0.0
    else:
        ts = torch.full((B, 1), 0.7)
    frac = sched.fractional_coords(f, h, w, 1, 25.0)
    return cfg, sd32, x, enc, mask, ts, frac


def build_model(cfg, sd32):
    import ltxmi
    m = ltxmi.Transformer3DModel(**cfg)
    m.load_state_dict(sd32)
    return m.to(device=DEV, dtype=BF).eval()


def run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid, truth_device="cpu", eager_device="cpu", **kw):
    """(fp32 truth, the reference's bf16 eager rendering) from the oracle's plain-PyTorch restatement.  ``*_device``: the
    oracle is device-agnostic torch code; for the sizes the host cannot turn around in a test (config 3 at full depth) it
    is run on the GPU -- rocBLAS / eager torch kernels, i.e. the reference's own eager path on this hardware -- and
    still shares nothing with libltxmi."""
    from oracle import dit

    def one(dtype, device):
        sd = {k: v.to(device=device, dtype=dtype) for k, v in sd32.items()}
        fc = tuple(t.to(device) for t in dit.precompute_freqs_cis(frac, cfg, dtype))
        k2 = dict(kw)
        if k2.get("skip_layer_mask") is not None:
            k2["skip_layer_mask"] = k2["skip_layer_mask"].to(device=device, dtype=dtype)
        out = dit.transformer3d_forward(sd, cfg, x.to(device=device, dtype=dtype), fc, enc.to(device=device, dtype=dtype),
                                        ts.to(device), encoder_attention_mask=mask.to(device), latent_shape=grid, **k2)
        return out.cpu()

    return one(torch.float32, truth_device), one(BF, eager_device)


class _Holder:
    _interrupt = False


@pytest.mark.parametrize("per_token", [False, True])
def test_transformer_small(per_token):
    """2 layers, D = 128 (2 heads x 64): every kernel on the path, ragged tile edges."""
    grid, B, T = (3, 5, 7), 3, 40
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 2, grid, B, T, per_token=per_token)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), latent_shape=grid, ltxv_model=_Holder(), return_dict=False)[0]
    assert_parity(out, truth, eager, f"small per_token={per_token}")


@pytest.mark.parametrize("strategy", ["AttentionValues", "AttentionSkip", "TransformerBlock"])
def test_transformer_stg_strategies(strategy):
    import ltxmi
    from oracle import dit
    grid, B, T = (2, 4, 8), 3, 24
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 3, grid, B, T, seed=3)
    skip = dit.create_skip_layer_mask(3, 1, 3, 2, [1], torch.float32)
    code = {"AttentionValues": dit.ATTENTION_VALUES, "AttentionSkip": dit.ATTENTION_SKIP,
            "TransformerBlock": dit.TRANSFORMER_BLOCK}[strategy]
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid, skip_layer_mask=skip,
                               skip_layer_strategy=code)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    dmask = m.create_skip_layer_mask(1, 3, 2, [1])
    assert torch.equal(dmask.float().cpu(), skip)
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), skip_layer_mask=dmask, skip_layer_strategy=getattr(ltxmi.SkipLayerStrategy, strategy),
            latent_shape=grid, ltxv_model=_Holder(), return_dict=False)[0]
    assert_parity(out, truth, eager, f"stg {strategy}")
    # the perturbed row must differ from the unperturbed one (the mask really took effect)
    assert rel(out[2], out[1]) > 1e-3


def test_transformer_joint_pass_false_matches_joint():
    import ltxmi
    grid, B, T = (2, 4, 8), 3, 24
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 2, grid, B, T, seed=4)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    dmask = m.create_skip_layer_mask(1, 3, 2, [0])
    kw = dict(freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
              timestep=ts.to(DEV), skip_layer_mask=dmask, skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues,
              latent_shape=grid, ltxv_model=_Holder(), return_dict=False)
    a = m(x.to(DEV), joint_pass=True, **kw)[0]
    b = m(x.to(DEV), joint_pass=False, **kw)[0]
    assert rel(b, a) < 4e-3


@pytest.mark.parametrize("alias", [0, 2])
def test_transformer_stacked_text_kv_is_bit_identical(alias):
    """Without the per-generation cache a forward projects the prompt through ALL layers' [to_k; to_v] in one stacked GEMM
    (ops.STACKED_TEXT_KV) instead of one skinny GEMM per block: same values bit for bit, with the full batch and with the
    leading rows a block sees under ``stg_alias_blocks``; a stale hand-over (other prompt tensor, edited weights) is not used."""
    from ltxmi import ops
    grid = (2, 4, 6)
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 4, grid, 3, 40, seed=5)
    x[2], enc[2], mask[2], ts[2] = x[1], enc[1], mask[1], ts[1]
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    encd = enc.to(DEV)
    kw = dict(freqs_cis=fc, encoder_attention_mask=mask.to(DEV), timestep=ts.to(DEV), latent_shape=grid,
              ltxv_model=_Holder(), return_dict=False, stg_alias_blocks=alias)
    old = (ops.STEP_INVARIANT_CACHING, ops.STACKED_TEXT_KV)
    try:
        ops.set_step_invariant_caching(False)
        ops.STACKED_TEXT_KV = False
        ref = m(x.to(DEV), encoder_hidden_states=encd, **kw)[0]
        ops.STACKED_TEXT_KV = True
        out = m(x.to(DEV), encoder_hidden_states=encd, **kw)[0]
        assert "_stacked_kv_weights" in m.__dict__                          # the stacked path ran ...
        assert all("_text_kv_ready" not in b.attn2.__dict__ for b in m.transformer_blocks)   # ... and every block took its slice
        assert torch.equal(out, ref)
        # another prompt tensor: new projection, not the previous forward's
        enc2 = (encd.float() * 0.5).to(encd.dtype)
        ops.STACKED_TEXT_KV = False
        ref2 = m(x.to(DEV), encoder_hidden_states=enc2, **kw)[0]
        ops.STACKED_TEXT_KV = True
        out2 = m(x.to(DEV), encoder_hidden_states=enc2, **kw)[0]
        assert torch.equal(out2, ref2) and not torch.equal(out2, out)
        # an in-place weight edit rebuilds the stack
        with torch.no_grad():
            m.transformer_blocks[1].attn2.to_k.weight.mul_(0.5)
        out3 = m(x.to(DEV), encoder_hidden_states=encd, **kw)[0]
        ops.STACKED_TEXT_KV = False
        ref3 = m(x.to(DEV), encoder_hidden_states=encd, **kw)[0]
        assert torch.equal(out3, ref3) and not torch.equal(out3, out)
    finally:
        ops.set_step_invariant_caching(old[0])
        ops.STACKED_TEXT_KV = old[1]


def test_transformer_stg_row_alias_is_bit_identical():
    """stg_alias_blocks: the perturbed row computed as a copy of the text row up to the first skipped block
    gives bit-identical output to computing all three rows."""
    grid = (2, 4, 6)
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 4, grid, 3, 16, seed=3)
    x[2] = x[1]
    enc[2] = enc[1]
    mask[2] = mask[1]
    ts[2] = ts[1]
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    from ltxmi import SkipLayerStrategy
    for strategy in (SkipLayerStrategy.AttentionValues, SkipLayerStrategy.TransformerBlock):
        slm = m.create_skip_layer_mask(1, 3, 2, [2, 3])
        kw = dict(freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
                  timestep=ts.to(DEV), skip_layer_mask=slm, skip_layer_strategy=strategy, latent_shape=grid,
                  return_dict=False)
        full = m(x.to(DEV).clone(), **kw)[0]
        dedup = m(x.to(DEV).clone(), stg_alias_blocks=2, **kw)[0]
        assert torch.equal(full, dedup)
        assert not torch.equal(full[1], full[2])            # the perturbation does act after block 2


@pytest.mark.parametrize("grid,note", [((4, 24, 32), "B and B-1 rows on the same (pipelined) attention kernel, GEMM tile choice changes with M"),
                                       ((4, 32, 32), "B rows on the pipelined attention kernel, B-1 rows not: aliasing is dropped")])
def test_transformer_stg_row_alias_across_kernel_thresholds(grid, note):
    """The same exactness where the dispatchers sit near their thresholds: 4 heads x 64, N = 3072 (the GEMM tile
    choice and tile positions of a row change with M; attention kernel ids equal) and
    N = 4096 (3 x 4 x 16 = 192 workgroups of 256 rows >= the pipelined kernel's threshold for B = 3, 128 for B = 2:
    different attention kernels)."""
    from ltxmi import SkipLayerStrategy, ops
    cfg, sd32, x, enc, mask, ts, frac = dit_case(4, 64, 2, grid, 3, 16, seed=13)
    x[2], enc[2], mask[2], ts[2] = x[1], enc[1], mask[1], ts[1]
    m = build_model(cfg, sd32)
    N = grid[0] * grid[1] * grid[2]
    same = ops.attention_kernel_id(3, 4, N, N, 64) == ops.attention_kernel_id(2, 4, N, N, 64)
    assert same == (N == 3072), note
    fc = m.precompute_freqs_cis(frac.to(DEV))
    slm = m.create_skip_layer_mask(1, 3, 2, [1])
    kw = dict(freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV), timestep=ts.to(DEV),
              skip_layer_mask=slm, skip_layer_strategy=SkipLayerStrategy.AttentionValues, latent_shape=grid, return_dict=False)
    full = m(x.to(DEV).clone(), **kw)[0]
    dedup = m(x.to(DEV).clone(), stg_alias_blocks=1, **kw)[0]
    assert torch.equal(full, dedup), note


def test_transformer_interrupt_and_output_types():
    import ltxmi
    grid, B, T = (2, 2, 4), 1, 16
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 1, grid, B, T, seed=5)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    kw = dict(freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
              timestep=ts.to(DEV), latent_shape=grid)
    out = m(x.to(DEV), ltxv_model=_Holder(), **kw)
    assert isinstance(out, ltxmi.Transformer3DModelOutput) and out.sample.shape == (B, 16, 128)

    class Stop:
        _interrupt = True
    assert m(x.to(DEV), ltxv_model=Stop(), **kw) == [None]                 # transformer3d.py:468-469


def test_transformer_2b_width_two_layers():
    """The real 2B widths (D 2048, 32 heads x 64, FF 8192, caption 4096), 2 layers, N = 1040 tokens
    (not a multiple of the 128/256 tiles), B_eff = 3 with the STG row."""
    import ltxmi
    from oracle import dit
    grid, B, T = (5, 13, 16), 3, 256
    cfg, sd32, x, enc, mask, ts, frac = dit_case(32, 64, 2, grid, B, T, caption=4096, seed=6)
    skip = dit.create_skip_layer_mask(2, 1, 3, 2, [1], torch.float32)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid, skip_layer_mask=skip,
                               skip_layer_strategy=dit.ATTENTION_VALUES)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [1]),
            skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=grid,
            ltxv_model=_Holder(), return_dict=False)[0]
    assert_parity(out, truth, eager, "2B-width, 2 layers, N 1040")


def test_transformer_2b_one_block_full_size():
    """Config 2 at FULL size: 2B widths, N = 4992 tokens (13 x 16 x 24), B_eff = 3 (CFG + STG rows), T = 256 --
    the tensors the bench times -- through ONE transformer block plus the embeddings and the output head, against
    the fp32 oracle and the reference's bf16 eager rendering of the same computation."""
    import ltxmi
    from oracle import dit
    grid, B, T = (13, 16, 24), 3, 256
    cfg, sd32, x, enc, mask, ts, frac = dit_case(32, 64, 1, grid, B, T, caption=4096, seed=16)
    skip = dit.create_skip_layer_mask(1, 1, 3, 2, [0], torch.float32)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid, skip_layer_mask=skip,
                               skip_layer_strategy=dit.ATTENTION_VALUES)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [0]),
            skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=grid,
            ltxv_model=_Holder(), return_dict=False)[0]
    assert out.shape == (3, 4992, 128)
    assert_parity(out, truth, eager, "2B-width, 1 block, N 4992, B_eff 3")


def test_transformer_13b_width_one_block():
    """The widths of the model the reference actually ships (LTXV is hard-wired to the 13B 0.9.7 checkpoint, ltxv.py:171-194):
    D = 4096 = 32 heads x 128, FF = 4 D = 16384, caption 4096 -- assumed from the published 13B architecture; the real
    config lives in the checkpoint's metadata, which the repo does not hold -- through ONE BasicTransformerBlock plus the
    embeddings and the output head.  N = 1040 tokens x B_eff 2 puts self-attention on the head_dim-128 pipelined kernel
    (2 x 32 x 5 = 320 workgroups, 1040 keys) with q's norm + RoPE applied on load over 64 partial sums per row."""
    import ltxmi
    from ltxmi import ops
    from oracle import dit
    grid, B, T = (5, 13, 16), 2, 128
    cfg, sd32, x, enc, mask, ts, frac = dit_case(32, 128, 1, grid, B, T, caption=4096, seed=26)
    assert cfg["num_attention_heads"] * cfg["attention_head_dim"] == 4096
    assert ops.attention_kernel_id(B, 32, 1040, 1040, 128) == 6
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid)
    m = build_model(cfg, sd32)
    assert m.transformer_blocks[0].ff.net[2].weight.shape == (4096, 16384)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), latent_shape=grid, ltxv_model=_Holder(), return_dict=False)[0]
    assert out.shape == (2, 1040, 128)
    assert_parity(out, truth, eager, "13B width (D 4096, 32 x 128), 1 block, N 1040")


def test_transformer_2b_full_depth():
    """The whole 2B model: all 28 layers at the real widths (D 2048, 32 x 64 heads, FF 8192, caption 4096, T 256), B_eff 3
    with the STG row perturbed from block 19 (the 2B default), on an eighth-size token grid (N = 624) so that the two CPU
    oracle runs (fp32 truth, bf16 eager) finish in about a minute.  Error accumulated through the full depth must stay
    within the rounding the reference's own bf16 eager run accumulates."""
    import ltxmi
    from oracle import dit
    grid, B, T = (2, 13, 24), 3, 256
    cfg, sd32, x, enc, mask, ts, frac = dit_case(32, 64, 28, grid, B, T, caption=4096, seed=26)
    skip = dit.create_skip_layer_mask(28, 1, 3, 2, [19], torch.float32)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid, skip_layer_mask=skip,
                               skip_layer_strategy=dit.ATTENTION_VALUES)
    m = build_model(cfg, sd32)
    del sd32
    fc = m.precompute_freqs_cis(frac.to(DEV))
    kw = dict(freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV), timestep=ts.to(DEV),
              skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [19]), skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues,
              latent_shape=grid, ltxv_model=_Holder(), return_dict=False)
    out = m(x.to(DEV), **kw)[0]
    assert out.shape == (3, 624, 128)
    assert_parity(out, truth, eager, "2B, 28 layers, N 624, B_eff 3")
    # and the pipeline's row de-duplication over the 19 blocks before the first skipped one: same bits
    x2 = x.clone()
    x2[2] = x2[1]
    enc2, mask2, ts2 = enc.clone(), mask.clone(), ts.clone()
    enc2[2], mask2[2], ts2[2] = enc2[1], mask2[1], ts2[1]
    kw.update(encoder_hidden_states=enc2.to(DEV), encoder_attention_mask=mask2.to(DEV), timestep=ts2.to(DEV))
    full = m(x2.to(DEV).clone(), **kw)[0]
    dedup = m(x2.to(DEV).clone(), stg_alias_blocks=19, **kw)[0]
    assert torch.equal(full, dedup)


CONFIG3_GRID = (16, 22, 38)           # BASELINE config 3: 1216 x 704 x 121 -> 13 376 latent tokens


def _config3_case(layers, seed):
    """B_eff 3 (CFG + STG rows), per-token timesteps with the first latent frame conditioned (t = 0 there: the i2v form,
    pipeline_ltx_video.py:1145-1150), T = 256 text tokens at the 2B widths."""
    cfg, sd32, x, enc, mask, ts, frac = dit_case(32, 64, layers, CONFIG3_GRID, 3, 256, caption=4096, seed=seed, per_token=True)
    assert x.shape == (3, 13376, 128) and ts.shape == (3, 13376) and float(ts[:, :22 * 38].max()) == 0.0
    return cfg, sd32, x, enc, mask, ts, frac


def test_transformer_config3_one_block_full_size():
    """Config 3 on ONE GPU, one block: N = 13 376 tokens x B_eff 3 with per-token (per-frame) timesteps through one
    BasicTransformerBlock + embeddings + output head (transformer3d.py:420-425 reshapes the timestep to [B, F], the AdaLN
    tables then differ per frame) against the fp32 oracle run on the host (~9 TFLOP) and the bf16 eager rendering (the
    oracle in bf16 on the device: eager torch kernels)."""
    import ltxmi
    from oracle import dit
    cfg, sd32, x, enc, mask, ts, frac = _config3_case(1, seed=36)
    skip = dit.create_skip_layer_mask(1, 1, 3, 2, [0], torch.float32)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, CONFIG3_GRID, eager_device=DEV, skip_layer_mask=skip,
                               skip_layer_strategy=dit.ATTENTION_VALUES)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [0]),
            skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=CONFIG3_GRID,
            ltxv_model=_Holder(), return_dict=False)[0]
    assert out.shape == (3, 13376, 128)
    assert_parity(out, truth, eager, "config 3: 2B width, 1 block, N 13376, B_eff 3, per-token timesteps")


def test_transformer_config3_full_depth_full_size():
    """Config 3 on ONE GPU, the whole model: 28 layers at N = 13 376 x B_eff 3, per-token timesteps, STG row perturbed from
    block 19 -- the workload ``bench.py``'s ``config3_step_1gpu`` leg times, and the N = 1 anchor of the 1 -> 8 curve.  The
    two oracle runs (fp32 truth, bf16 eager: ~0.5 PFLOP each) are far beyond the host, so the oracle's PyTorch code runs
    on the device in fp32 / bf16 (rocBLAS + eager kernels); the bound is the usual one: our error against the fp32 result
    may not exceed the eager bf16 path's by more than 2e-3."""
    import ltxmi
    from oracle import dit
    cfg, sd32, x, enc, mask, ts, frac = _config3_case(28, seed=46)
    skip = dit.create_skip_layer_mask(28, 1, 3, 2, [19], torch.float32)
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, CONFIG3_GRID, truth_device=DEV, eager_device=DEV,
                               skip_layer_mask=skip, skip_layer_strategy=dit.ATTENTION_VALUES)
    torch.cuda.empty_cache()
    m = build_model(cfg, sd32)
    del sd32
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [19]),
            skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=CONFIG3_GRID,
            ltxv_model=_Holder(), return_dict=False)[0]
    assert out.shape == (3, 13376, 128)
    assert_parity(out, truth, eager, "config 3: 2B, 28 layers, N 13376, B_eff 3, per-token timesteps")


# ------------------------------------------------------------------------------- VAE
def vae_case(style, base=64, latent=128, seed=0, with_encoder=False):
    from oracle import vae as ov
    from oracle import vae_encoder as oe
    if style == "b":
        cfg = ov.demo_config(latent)
    else:
        cfg = {"_class_name": "CausalVideoAutoencoder", "dims": 3, "in_channels": 3, "out_channels": 3,
               "latent_channels": latent,
               "blocks": [["res_x", 1], ["compress_all", 1], ["res_x_y", 1], ["res_x", 1], ["compress_all", 1],
                          ["res_x_y", 1], ["res_x", 1], ["compress_all", 1], ["res_x", 1], ["res_x", 1]],
               "scaling_factor": 1.0, "norm_layer": "pixel_norm", "patch_size": 4, "latent_log_var": "uniform",
               "use_quant_conv": False, "causal_decoder": False}
    cfg["decoder_base_channels"] = base
    cfg["build_encoder"] = with_encoder
    raw = dict(ov.init_state_dict(cfg, seed=seed))
    if with_encoder:
        cfg["encoder_base_channels"] = base
        if style == "b":
            cfg["encoder_blocks"] = oe.demo_encoder_blocks()
        raw.update(oe.init_state_dict(cfg, seed=seed + 1))
    sd = {k: (v.to(BF).float() if v.is_floating_point() and v.dim() > 0 else v) for k, v in raw.items()}
    return cfg, sd


def build_vae(cfg, sd):
    import ltxmi
    v = ltxmi.CausalVideoAutoencoder.from_config(dict(cfg))
    v.load_state_dict(sd)
    v = v.to(device=DEV, dtype=BF).eval()
    if cfg.get("timestep_conditioning"):
        v.decoder.timestep_scale_multiplier.data = v.decoder.timestep_scale_multiplier.data.float()
    return v


@pytest.mark.parametrize("style", ["a", "b"])
def test_vae_decode(style):
    from oracle import vae as ov
    import ltxmi
    cfg, sd = vae_case(style)
    z = torch.randn(1, 128, 3, 4, 5, generator=torch.Generator().manual_seed(9)).to(BF)
    ts = torch.tensor([0.05]) if cfg.get("timestep_conditioning") else None
    truth = ov.vae_decode(sd, cfg, z.float(), ts)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts)
    v = build_vae(cfg, sd)
    out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True,
                           timestep=None if ts is None else ts.to(DEV))
    assert out.shape == truth.shape == (1, 3, 17, 128, 160)
    assert_parity(out, truth, eager, f"vae {style}")
    # decode() contract: asserts on target_shape, DecoderOutput vs tuple (vae.py:357-364,410-413)
    with pytest.raises(AssertionError):
        v.decode(z.to(DEV))
    o = v.decode(z.to(DEV), target_shape=truth.shape, timestep=None if ts is None else ts.to(DEV))
    assert isinstance(o, ltxmi.DecoderOutput)


def test_vae_decode_through_direct_convolution():
    """A latent large enough (9 x 256 x 256 pixels) that the full-resolution C = 128 layers and conv_out take the
    direct-convolution kernel (>= 128 workgroups) inside the real decoder."""
    from oracle import vae as ov
    import ltxmi
    cfg, sd = vae_case("b")
    z = torch.randn(1, 128, 2, 8, 8, generator=torch.Generator().manual_seed(15)).to(BF)
    ts = torch.tensor([0.05])
    truth = ov.vae_decode(sd, cfg, z.float(), ts)
    v = build_vae(cfg, sd)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts)
    out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    assert out.shape == truth.shape == (1, 3, 9, 256, 256)
    assert_parity(out, truth, eager, "vae decode through the direct convolution")


@pytest.mark.parametrize("knobs", [(False, False, False), (True, False, False), (True, True, False), (True, True, True)])
def test_vae_decode_with_the_epilogue_fusions_switched_off(knobs, monkeypatch):
    """The decoder hands every PixelNorm -> AdaLN -> SiLU to the producer of its input; what the producer does with it is three
    switches in ltxmi.ops (norm2 in conv1's epilogue, the consumer's norm as a second output, the channel split of the wide
    stages with its finalising pass) and, for a C caller, the optional fields y_norm / workspace.  Every combination -- down to
    every norm as a launch of its own, which is what a caller that passes none of the optional fields gets -- must render the
    same video within the parity bound of the fp32 oracle, at the full-width decoder's channel counts."""
    from oracle import vae as ov
    from ltxmi import ops
    import ltxmi
    cfg, sd = vae_case("b", base=128)
    z = torch.randn(1, 128, 3, 6, 8, generator=torch.Generator().manual_seed(19)).to(BF)
    ts = torch.tensor([0.05])
    truth = ov.vae_decode(sd, cfg, z.float(), ts)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts)
    v = build_vae(cfg, sd)
    monkeypatch.setattr(ops, "CONV_POST_NORM_FUSE", knobs[0])
    monkeypatch.setattr(ops, "CONV_SECOND_OUTPUT_FUSE", knobs[1])
    monkeypatch.setattr(ops, "CONV_SPLIT", knobs[2])
    out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    assert_parity(out, truth, eager, f"vae decode, base 128, fusions {knobs}")


def test_vae_decode_full_width():
    """The decoder the bench times: decoder_base_channels = 128 (1024 / 512 / 256 / 128 channels per stage, the
    d2s 1024 -> 4096 upsampler), on a latent the CPU oracle can decode in seconds."""
    from oracle import vae as ov
    import ltxmi
    cfg, sd = vae_case("b", base=128)
    z = torch.randn(1, 128, 3, 6, 8, generator=torch.Generator().manual_seed(17)).to(BF)
    ts = torch.tensor([0.05])
    truth = ov.vae_decode(sd, cfg, z.float(), ts)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts)
    v = build_vae(cfg, sd)
    out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    assert out.shape == truth.shape == (1, 3, 17, 192, 256)
    assert_parity(out, truth, eager, "vae decode, base 128")


def _vae_small_eager_error(cfg, sd):
    """What the reference's bf16 eager decode loses against fp32 with THESE weights, measured where the CPU can run the
    bf16 oracle (the latent of test_vae_decode_full_width): (rel L2, max err / range).  The full-size tests below bound
    the product's error by it (+ BASELINE's 2e-3) -- the per-element error of a decode does not depend on how many
    positions there are."""
    from oracle import vae as ov
    z = torch.randn(1, 128, 3, 6, 8, generator=torch.Generator().manual_seed(17)).to(BF)
    ts = torch.tensor([0.05])
    truth = ov.vae_decode(sd, cfg, z.float(), ts)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts)
    return rel(eager, truth), maxrel(eager, truth)


def _assert_full_size_parity(out, truth, e_ref, m_ref, what):
    e, m = rel(out, truth), maxrel(out, truth)
    print(f"{what}: rel L2 {e:.3e} (bound: bf16 eager {e_ref:.3e} + {RTOL});  max err / range {m:.3e} (eager {m_ref:.3e})")
    assert torch.isfinite(out.float()).all(), what
    assert e <= e_ref + RTOL, (what, e, e_ref)
    assert m <= 1.5 * m_ref + 1e-2, (what, m, m_ref)
    return e


def _cpu_threads():
    import os
    n = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(n)
    return n


def test_vae_decode_config2_full_size_vs_fp32_oracle():
    """BASELINE config 2's decode at FULL size -- z [1,128,13,16,24] -> 97 x 512 x 768, decoder_base_channels 128, the
    tensors the bench times -- against the fp32 CPU oracle (24.4 TFLOP on the host: about a minute on 16 threads), for
    BOTH convolution paths: chosen by shape (the direct kernel wherever it applies) and all-implicit-GEMM.  Round 2 only
    compared the two with each other."""
    import time
    from oracle import vae as ov
    import ltxmi
    from ltxmi import ops
    cfg, sd = vae_case("b", base=128)
    e_ref, m_ref = _vae_small_eager_error(cfg, sd)
    v = build_vae(cfg, sd)
    z = torch.randn(1, 128, 13, 16, 24, generator=torch.Generator().manual_seed(18)).to(BF)
    ts = torch.tensor([0.05])
    out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    assert out.shape == (1, 3, 97, 512, 768)
    old = ops.CONV_ALGO
    try:
        ops.CONV_ALGO = 1
        out_ig = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    finally:
        ops.CONV_ALGO = old
    out, out_ig = out.cpu(), out_ig.cpu()
    n = _cpu_threads()
    t0 = time.time()
    with torch.no_grad():
        truth = ov.vae_decode(sd, cfg, z.float(), ts)
    print(f"fp32 oracle decode of config 2 on {n} host threads: {time.time() - t0:.1f} s")
    assert truth.shape == out.shape
    e_d = _assert_full_size_parity(out, truth, e_ref, m_ref, "config-2 decode, convolutions by shape")
    e_i = _assert_full_size_parity(out_ig, truth, e_ref, m_ref, "config-2 decode, all implicit GEMM")
    # the two renderings against each other: bounded by the triangle inequality from the two oracle distances (round 2's
    # 8e-3 was a guess that the measured 9.7e-3 broke: two independent renderings ~8e-3 from the truth sit ~sqrt(2) x
    # that apart).  Kept as a report + the exact bound.
    e_x = rel(out, out_ig)
    print(f"config-2 decode, direct vs implicit GEMM: rel L2 {e_x:.3e} (oracle distances {e_d:.3e} / {e_i:.3e})")
    assert e_x <= (e_d + e_i) * 1.05 + 1e-4


def test_vae_decode_config5_full_size_tiled():
    """BASELINE config 5 at FULL size: z [1,128,33,23,40] -> 257 frames of 736 x 1280, z-tiled by 4 latent frames (11
    tiles, vae.py:365-402).  (a) tile 0 alone (5 latent frames, 22 TFLOP) against the fp32 CPU oracle; (b) the tiled
    output's frames before the first blend are tile 0's untiled decode bit for bit; (c) every later tile's unblended
    middle frames equal that tile's own untiled decode; (d) the whole tiled output with all-implicit-GEMM convolutions
    against the one with convolutions chosen by shape."""
    import time
    from oracle import vae as ov
    import ltxmi
    from ltxmi import ops
    cfg, sd = vae_case("b", base=128)
    e_ref, m_ref = _vae_small_eager_error(cfg, sd)
    v = build_vae(cfg, sd)
    z = torch.randn(1, 128, 33, 23, 40, generator=torch.Generator().manual_seed(19)).to(BF)
    ts = torch.tensor([0.05])
    zd, tsd = z.to(DEV), ts.to(DEV)
    v.enable_z_tiling(4)
    tiled = ltxmi.vae_decode(zd, v, True, vae_per_channel_normalize=True, timestep=tsd)
    assert tiled.shape == (1, 3, 257, 736, 1280) and tiled.dtype == torch.float16 and torch.isfinite(tiled.float()).all()
    old = ops.CONV_ALGO
    try:
        ops.CONV_ALGO = 1
        tiled_ig = ltxmi.vae_decode(zd, v, True, vae_per_channel_normalize=True, timestep=tsd)
    finally:
        ops.CONV_ALGO = old
    v.disable_z_tiling()
    # (b) frames 0 .. 24 of the tiled output = tile 0 (latent frames 0 .. 4) decoded on its own, in fp16
    tile0 = ltxmi.vae_decode(zd[:, :, 0:5], v, True, vae_per_channel_normalize=True, timestep=tsd)
    assert tile0.shape == (1, 3, 33, 736, 1280)
    assert torch.equal(tiled[:, :, :25], tile0.to(torch.float16)[:, :, :25])
    # (c) tile n (latent frames 3n .. 3n + 4) contributes output frames 25 + 24 (n - 1) ... ; its first 8 are cross-faded
    # with tile n - 1, the next 16 are its own frames 9 .. 24 (frame 0 dropped) untouched
    for n in (1, 5, 10):
        zt = zd[:, :, 3 * n:3 * n + 5]
        own = ltxmi.vae_decode(zt, v, True, vae_per_channel_normalize=True, timestep=tsd).to(torch.float16)[:, :, 1:]
        lo = 25 + 24 * (n - 1)
        keep = min(24, own.shape[2]) - 8
        assert torch.equal(tiled[:, :, lo + 8:lo + 8 + keep], own[:, :, 8:8 + keep]), f"tile {n}"
    # (d) the two convolution paths over the whole tiled clip (bound: twice the oracle distance allowed per path)
    e_x = rel(tiled, tiled_ig)
    print(f"config-5 tiled decode, direct vs implicit GEMM: rel L2 {e_x:.3e}")
    assert e_x <= 2 * (e_ref + RTOL)
    del tiled_ig
    # (a) tile 0 against the fp32 oracle
    n = _cpu_threads()
    t0 = time.time()
    with torch.no_grad():
        truth = ov.vae_decode(sd, cfg, z[:, :, 0:5].float(), ts)
    print(f"fp32 oracle decode of config 5's tile 0 on {n} host threads: {time.time() - t0:.1f} s")
    _assert_full_size_parity(tile0.cpu(), truth, e_ref, m_ref, "config-5 tile 0 (5 x 23 x 40 latents)")


def test_vae_decoder_block_variants():
    """compress_time / compress_space upsamplers and inject_noise resnet blocks (causal_video_autoencoder.py:671-684,
    1183-1195; in no shipped config, pinned by golden G10c) in the product's decoder against the oracle fed the same
    noise draws (a twin of the device generator, drawn in the reference's order)."""
    from oracle import vae as ov
    import ltxmi
    from ltxmi import autoencoder as ae
    cfg = {"_class_name": "CausalVideoAutoencoder", "dims": 3, "in_channels": 3, "out_channels": 3, "latent_channels": 128,
           "decoder_blocks": [["res_x", {"num_layers": 1}], ["compress_space", {}],
                              ["res_x", {"num_layers": 2, "inject_noise": True}], ["compress_time", {}],
                              ["res_x", {"num_layers": 1, "inject_noise": True}]],
           "scaling_factor": 1.0, "norm_layer": "pixel_norm", "patch_size": 4, "latent_log_var": "uniform",
           "use_quant_conv": False, "causal_decoder": False, "timestep_conditioning": True,
           "spatial_padding_mode": "replicate", "decoder_base_channels": 64, "build_encoder": False}
    raw = dict(ov.init_state_dict(cfg, seed=21))
    sd = {k: (v.to(BF).float() if v.is_floating_point() and v.dim() > 0 else v) for k, v in raw.items()}
    assert any("per_channel_scale1" in k for k in sd)
    z = torch.randn(1, 128, 3, 4, 5, generator=torch.Generator().manual_seed(22)).to(BF)
    ts = torch.tensor([0.05])
    twin = torch.Generator(device=DEV).manual_seed(23)
    noises = [torch.randn((4, 5), generator=twin, device=DEV, dtype=BF).float().cpu() for _ in range(6)]
    truth = ov.vae_decode(sd, cfg, z.float(), ts, noises=noises)
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = ov.vae_decode(sdb, cfg, z, ts, noises=[n.to(BF) for n in noises])
    v = build_vae(cfg, sd)
    old = ae.ResnetBlock3D.noise_generator
    try:
        ae.ResnetBlock3D.noise_generator = torch.Generator(device=DEV).manual_seed(23)
        out = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=ts.to(DEV))
    finally:
        ae.ResnetBlock3D.noise_generator = old
    assert out.shape == truth.shape == (1, 3, 5, 32, 40)
    assert_parity(out, truth, eager, "decoder block variants (compress_time / compress_space / inject_noise)")
    with pytest.raises(NotImplementedError, match="unreachable in the reference"):
        ltxmi.CausalVideoAutoencoder.from_config(dict(cfg, decoder_blocks=[["attn_res_x", {"num_layers": 1, "attention_head_dim": 32}]]))


def test_checkpoint_loaded_from_disk_runs_the_oracles_forward(tmp_path):
    """f4 on the device: a single-file safetensors checkpoint with the reference's key prefixes and the config blob in
    its metadata (transformer3d.py:271-326, causal_video_autoencoder.py:35-120), written to disk, loaded with
    from_pretrained and RUN -- DiT forward and VAE decode against the oracle on the same weights."""
    import json
    import os
    from safetensors.torch import save_file
    from oracle import dit, vae as ov
    import ltxmi
    grid, B, T = (2, 4, 6), 2, 24
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 2, grid, B, T, seed=31)
    vcfg, vsd = vae_case("b")
    blob = {"model.diffusion_model." + k: v.to(BF) for k, v in sd32.items()}
    blob.update({"vae." + k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in vsd.items()})
    path = os.path.join(tmp_path, "ltxv.safetensors")
    save_file(blob, path, metadata={"config": json.dumps({"transformer": cfg, "vae": vcfg})})
    m = ltxmi.Transformer3DModel.from_pretrained(path, device=DEV).eval()
    truth, eager = run_oracles(cfg, sd32, x, enc, mask, ts, frac, grid)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    out = m(x.to(DEV), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
            timestep=ts.to(DEV), latent_shape=grid, ltxv_model=_Holder(), return_dict=False)[0]
    assert_parity(out, truth, eager, "DiT loaded from a single-file checkpoint")
    v = ltxmi.CausalVideoAutoencoder.from_pretrained(path, device=DEV).eval()
    if vcfg.get("timestep_conditioning"):
        v.decoder.timestep_scale_multiplier.data = v.decoder.timestep_scale_multiplier.data.float()
    z = torch.randn(1, 128, 2, 3, 4, generator=torch.Generator().manual_seed(32)).to(BF)
    t05 = torch.tensor([0.05])
    vtruth = ov.vae_decode(vsd, vcfg, z.float(), t05)
    veager = ov.vae_decode({k: (a.to(BF) if a.is_floating_point() and a.dim() > 0 else a) for k, a in vsd.items()}, vcfg, z, t05)
    vout = ltxmi.vae_decode(z.to(DEV), v, True, vae_per_channel_normalize=True, timestep=t05.to(DEV))
    assert_parity(vout, vtruth, veager, "VAE loaded from a single-file checkpoint")


def test_vae_tiled_decode_matches_oracle_tiling():
    from oracle import vae as ov
    cfg, sd = vae_case("b", base=64)
    ts = torch.tensor([0.05])
    v = build_vae(cfg, sd)
    # z-tiling (tile = 4+1 latent frames)
    z = torch.randn(1, 128, 7, 2, 2, generator=torch.Generator().manual_seed(10)).to(BF)
    sdb = {k: (x.to(BF) if x.is_floating_point() and x.dim() > 0 else x) for k, x in sd.items()}
    truth = ov.decode(sd, cfg, z.float(), ts, use_z_tiling=True, z_sample_size=4).float()
    eager = ov.decode(sdb, cfg, z, ts, use_z_tiling=True, z_sample_size=4).float()
    v.enable_z_tiling(4)
    out = v.decode(z.to(DEV), return_dict=False, target_shape=(1, 3, 49, 64, 64), timestep=ts.to(DEV))[0]
    v.disable_z_tiling()
    assert out.dtype == torch.float16 and out.shape == truth.shape
    assert_parity(out, truth, eager, "z-tiled decode")
    # hw-tiling with 64-px tiles
    z = torch.randn(1, 128, 2, 3, 4, generator=torch.Generator().manual_seed(11)).to(BF)
    truth = ov.decode(sd, cfg, z.float(), ts, use_hw_tiling=True, tile_sample_min_size=64)
    eager = ov.decode(sdb, cfg, z, ts, use_hw_tiling=True, tile_sample_min_size=64)
    v.set_tiling_params(sample_size=64, overlap_factor=0.25)
    v.enable_hw_tiling()
    out = v.decode(z.to(DEV), return_dict=False, target_shape=(1, 3, 9, 96, 128), timestep=ts.to(DEV))[0]
    v.disable_hw_tiling()
    assert out.shape == truth.shape
    assert_parity(out, truth, eager, "hw-tiled decode")


@pytest.mark.parametrize("style", ["a", "b"])
def test_vae_encode(style):
    """Encoder.forward + encode + vae_encode (conditioning path) vs the oracle, both block plans
    (a: strided compress_all + res_x_y; b: space-to-depth residual downsamples)."""
    from oracle import vae_encoder as oe
    import ltxmi
    cfg, sd = vae_case(style, with_encoder=True)
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(1, 3, 9, 64, 96, generator=g) * 2 - 1).to(BF)
    truth = oe.encode(sd, cfg, x.float())                                    # moments [1, 256, 2, 2, 3]
    sdb = {k: (v.to(BF) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    eager = oe.encode(sdb, cfg, x)
    v = build_vae(cfg, sd)
    dist = v.encode(x.to(DEV)).latent_dist
    assert isinstance(dist, ltxmi.DiagonalGaussianDistribution)
    out = dist.parameters
    assert out.shape == truth.shape == (1, 256, 2, 2, 3)
    e_ours, e_ref = rel(out, truth), rel(eager, truth)
    print(f"vae encode {style}: ours {e_ours:.3e}  reference-bf16-eager {e_ref:.3e}")
    assert torch.isfinite(out).all()
    assert e_ours <= e_ref + RTOL, (e_ours, e_ref)
    # vae_encode: mode (no sampling) == normalised mean; sampling adds std/s * noise
    want = oe.vae_encode(sd, cfg, x.float())
    got = ltxmi.vae_encode(x.to(DEV), v, vae_per_channel_normalize=True, sample_posterior=False)
    assert got.dtype == BF and rel(got, want) <= e_ref + 2 * RTOL
    gen = torch.Generator(device=DEV).manual_seed(3)
    noise = torch.randn(want.shape, generator=torch.Generator(device=DEV).manual_seed(3), device=DEV)
    want_s = oe.vae_encode(sd, cfg, x.float(), noise=noise.cpu())
    got_s = ltxmi.vae_encode(x.to(DEV), v, vae_per_channel_normalize=True, generator=gen)
    assert rel(got_s, want_s) <= e_ref + 2 * RTOL
    # a single image (F = 1) goes through the same causal path
    truth1 = oe.encode(sd, cfg, x[:, :, :1].float())
    out1 = v.encode(x[:, :, :1].to(DEV)).latent_dist.parameters
    assert out1.shape == truth1.shape and rel(out1, truth1) <= rel(oe.encode(sdb, cfg, x[:, :, :1]), truth1) + RTOL


def test_vae_tiled_encode_matches_oracle_tiling():
    from oracle import vae_encoder as oe
    cfg, sd = vae_case("b", with_encoder=True)
    v = build_vae(cfg, sd)
    g = torch.Generator().manual_seed(13)
    x = (torch.rand(1, 3, 41, 32, 32, generator=g) * 2 - 1).to(BF)
    truth = oe.encode(sd, cfg, x.float(), use_z_tiling=True, z_sample_size=4)
    v.enable_z_tiling(4)
    out = v.encode(x.to(DEV)).latent_dist.parameters
    v.disable_z_tiling()
    assert out.shape == truth.shape and rel(out, truth) < 2e-2
    x = (torch.rand(1, 3, 9, 128, 160, generator=g) * 2 - 1).to(BF)
    truth = oe.encode(sd, cfg, x.float(), use_hw_tiling=True, tile_sample_min_size=128)
    v.set_tiling_params(sample_size=128, overlap_factor=0.25)
    v.enable_hw_tiling()
    out = v.encode(x.to(DEV)).latent_dist.parameters
    v.disable_hw_tiling()
    assert out.shape == truth.shape and rel(out, truth) < 2e-2


@pytest.mark.parametrize("dims,mid", [(3, 64), (2, 128)])
def test_latent_upsampler_and_bridge(dims, mid):
    """LatentUpsampler.forward, _upsample_latents and adain_filter_latent (latent_upsampler.py:109-149,
    pipeline_ltx_video.py:1709-1737, 1760-1772) against the oracle; spatial upsampler, dims 3 (shipped) and 2."""
    import ltxmi
    from oracle import upsampler as ou
    cfg = dict(in_channels=128, mid_channels=mid, num_blocks_per_stage=2, dims=dims, spatial_upsample=True,
               temporal_upsample=False)
    sd = {k: v.to(BF).float() for k, v in ou.init_state_dict(cfg, seed=5).items()}
    g = torch.Generator().manual_seed(14)
    z = torch.randn(1, 128, 3, 6, 8, generator=g).to(BF)
    stats = {"per_channel_statistics.std-of-means": 0.5 + torch.rand(128, generator=g),
             "per_channel_statistics.mean-of-means": 0.2 * torch.randn(128, generator=g)}
    truth = ou.latent_upsampler_forward(sd, cfg, z.float())
    eager = ou.latent_upsampler_forward({k: v.to(BF) for k, v in sd.items()}, cfg, z)
    m = ltxmi.LatentUpsampler.from_config(cfg)
    m.load_state_dict(sd)
    m = m.to(device=DEV, dtype=BF).eval()
    assert m.config()["mid_channels"] == mid
    out = m(z.to(DEV))
    assert out.shape == truth.shape == (1, 128, 3, 12, 16)
    e_ours, e_ref = rel(out, truth), rel(eager, truth)
    print(f"latent upsampler dims={dims}: ours {e_ours:.3e}  reference-bf16-eager {e_ref:.3e}")
    assert e_ours <= e_ref + RTOL, (e_ours, e_ref)
    vae = types.SimpleNamespace(std_of_means=stats["per_channel_statistics.std-of-means"].to(DEV),
                                mean_of_means=stats["per_channel_statistics.mean-of-means"].to(DEV))
    want = ou.upsample_latents(sd, cfg, z.float(), stats)
    got = ltxmi.upsample_latents(m, z.float().to(DEV), vae)
    assert got.dtype == torch.float32 and rel(got, want) <= e_ref + 2 * RTOL
    want = ou.adain_filter_latent(want, z.float())
    got = ltxmi.adain_filter_latent(got, z.float().to(DEV))
    assert rel(got, want) <= e_ref + 2 * RTOL


# ------------------------------------------------------------------------ denoise loop
def _oracle_loop(sd32, cfg, noise_tokens, emb, msk, tsch, grid, gs, stg, rs, skips, dtype):
    """The oracle's restatement of the loop of LTXVideoPipeline.__call__ (:1104-1256), pinned to the reference's own
    __call__ by tests/test_oracle_golden.py::test_g11_config1_loop; dtype bf16 = the reference's eager rendering."""
    from oracle import dit, sched
    f, h, w = grid
    sd = {k: v.to(dtype) for k, v in sd32.items()}
    pix = sched.latent_to_pixel_coords(sched.get_latent_coords(f, h, w, 1),
                                       causal_fix=cfg.get("causal_temporal_positioning", False)).to(torch.float32)
    pix[:, 0] = pix[:, 0] * (1.0 / 25.0)
    fc = dit.precompute_freqs_cis(pix, cfg, dtype)
    lat = noise_tokens.clone().float()
    do_cfg, do_stg, do_rs = any(x > 1.0 for x in gs), any(x > 0.0 for x in stg), any(x != 1.0 for x in rs)
    n = 1 + int(do_cfg) + int(do_stg)
    for i, t in enumerate(tsch):
        skip = dit.create_skip_layer_mask(cfg["num_layers"], 1, n, n - 1, skips[i], dtype)
        npred = dit.transformer3d_forward(sd, cfg, torch.cat([lat.to(dtype)] * n), fc, emb.to(dtype), t.expand(n).unsqueeze(-1),
                                          encoder_attention_mask=msk, latent_shape=(f, h, w), skip_layer_mask=skip,
                                          skip_layer_strategy=dit.ATTENTION_VALUES)
        v = sched.guidance(npred.float(), n, gs[i], stg[i], rs[i], do_cfg, do_stg, do_rs)
        lat = sched.denoising_step(tsch, lat, v, t.expand(1).unsqueeze(-1), None, t)
    return sched.unpatchify(lat, f, h, w)


def test_pipeline_config1_two_steps():
    """BASELINE.json configs[0]: 256x256x9, 2 denoise steps -- the plumbing case.  Device loop
    (bf16 model, fp32 latents) against the oracle's fp32 loop on the same noise."""
    import ltxmi
    from oracle import dit, sched
    heads, dh, layers, caption, T = 2, 64, 2, 128, 32
    cfg = dict(dit.default_2b_config(), num_attention_heads=heads, attention_head_dim=dh, num_layers=layers,
               cross_attention_dim=heads * dh, caption_channels=caption)
    sd32 = {k: v.to(BF).float() for k, v in dit.init_state_dict(cfg, seed=7).items()}
    g = torch.Generator().manual_seed(8)
    f, h, w = 2, 8, 8
    N = f * h * w
    lat0 = torch.randn(1, N, 128, generator=g)
    pos, neg = torch.randn(1, T, caption, generator=g).to(BF), torch.randn(1, T, caption, generator=g).to(BF)
    pmask, nmask = torch.ones(1, T), torch.ones(1, T)
    pmask[:, 20:] = 0
    nmask[:, 5:] = 0
    gs, stg, rs, skip_blocks, steps = 3.0, 1.0, 0.7, [1], 2

    # ---- oracle loop, fp32 (pipeline_ltx_video.py:1104-1256)
    tsch = sched.set_timesteps(steps, (1, 128, f, h, w))
    emb = torch.cat([neg, pos, pos]).float()
    msk = torch.cat([nmask, pmask, pmask])
    truth = _oracle_loop(sd32, cfg, lat0, emb, msk, tsch, (f, h, w), [gs] * steps, [stg] * steps, [rs] * steps,
                         [skip_blocks] * steps, torch.float32)
    eager = _oracle_loop(sd32, cfg, lat0, emb, msk, tsch, (f, h, w), [gs] * steps, [stg] * steps, [rs] * steps,
                         [skip_blocks] * steps, BF)

    m = build_model(cfg, sd32)
    pipe = ltxmi.LTXVideoPipeline(transformer=m, scheduler=ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1))
    out = pipe(height=256, width=256, num_frames=9, frame_rate=25.0, prompt_embeds=pos.to(DEV),
               prompt_attention_mask=pmask.to(DEV),
               negative_prompt_embeds=neg.to(DEV), negative_prompt_attention_mask=nmask.to(DEV),
               num_inference_steps=steps, guidance_scale=gs, stg_scale=stg, rescaling_scale=rs,
               skip_block_list=skip_blocks, latents=lat0.to(DEV), output_type="latent",
               skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, is_video=True, joint_pass=True, latents_dtype=torch.float32)
    assert out.shape == truth.shape == (1, 128, f, h, w)
    torch.testing.assert_close(torch.tensor(pipe.scheduler.host_timesteps), tsch, rtol=1e-6, atol=1e-7)
    assert_parity(out, truth, eager, "pipeline config 1, 2 steps")


@pytest.mark.parametrize("run", ["", "tables."])
def test_pipeline_matches_the_references_own_call(golden, run):
    """G11 on the PRODUCT: ltxmi.LTXVideoPipeline on the weights, prompts and noise draw of the reference's own
    LTXVideoPipeline.__call__ run (tests/golden/g7_pipeline_call: config 1, 256x256x9, fp32 on the CPU) must land on the
    reference's output latents -- within what the reference's bf16 eager path (the oracle loop in bf16) manages.
    run "tables.": 3 steps with list-valued scales, guidance_timesteps and per-step skip lists (:959-1013)."""
    import ltxmi
    from oracle import pipeline_ctl as pc
    t, meta = golden("g7_pipeline_call")
    cfg = meta["cfg"]
    kw = dict(meta["tables_kwargs" if run else "kwargs"])
    f, h, w = meta["grid"]
    sd32 = {k[2:]: v for k, v in t.items() if k.startswith("w.")}
    truth = t[run + "out_latents"]
    ts = t[run + "timesteps"]
    emb = torch.cat([t["negative_prompt_embeds"], t["prompt_embeds"], t["prompt_embeds"]])
    msk = torch.cat([t["negative_prompt_attention_mask"], t["prompt_attention_mask"], t["prompt_attention_mask"]])
    gs, stg, rs, skips, _, _, _ = pc.guidance_tables(ts.tolist(), kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"],
                                                     kw["skip_block_list"], guidance_timesteps=kw.get("guidance_timesteps"))
    eager = _oracle_loop(sd32, cfg, t[run + "noise"], emb, msk, ts, (f, h, w), gs, stg, rs, skips, BF)
    m = build_model(cfg, sd32)
    pipe = ltxmi.LTXVideoPipeline(transformer=m, scheduler=ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1))
    kw["skip_layer_strategy"] = ltxmi.SkipLayerStrategy[kw["skip_layer_strategy"]]
    # exactly the keyword arguments the reference's own __call__ was run with (oracle/gen/make_golden.py g7), plus the
    # recorded noise draw in place of the generator and fp32 latents (the reference ran in fp32 throughout)
    out = pipe(prompt_embeds=t["prompt_embeds"].to(BF).to(DEV), prompt_attention_mask=t["prompt_attention_mask"].to(DEV),
               negative_prompt_embeds=t["negative_prompt_embeds"].to(BF).to(DEV),
               negative_prompt_attention_mask=t["negative_prompt_attention_mask"].to(DEV),
               latents=t[run + "noise"].to(DEV), output_type="latent", return_dict=False, is_video=True,
               vae_per_channel_normalize=True, joint_pass=True, ltxv_model=types.SimpleNamespace(_interrupt=False),
               latents_dtype=torch.float32, **kw)[0]
    torch.testing.assert_close(torch.tensor(pipe.scheduler.host_timesteps), ts, rtol=1e-6, atol=1e-7)
    assert out.shape == truth.shape
    assert_parity(out, truth, eager, f"product pipeline vs the reference's own __call__ ({run or 'config 1'})")


def test_prepare_conditioning_matches_oracle():
    """Token assembly for conditioning items (pipeline_ltx_video.py:1344-1548): first-frame image,
    mid-video sequence (prefix latents become extra tokens) and a single later frame."""
    import ltxmi
    from oracle import conditioning as oc, vae_encoder as oe
    cfg, sd = vae_case("b", with_encoder=True)
    v = build_vae(cfg, sd)
    pipe = ltxmi.LTXVideoPipeline(transformer=types.SimpleNamespace(config=types.SimpleNamespace(causal_temporal_positioning=True)),
                                  vae=v)
    H, W, F_ = 64, 96, 33
    g = torch.Generator().manual_seed(30)
    img, seq, single = [(torch.rand(1, 3, n, H, W, generator=g) * 2 - 1).to(BF) for n in (1, 17, 1)]
    init = torch.randn(1, 128, 5, 2, 3, generator=g)
    spec = [(img, 0, 1.0), (seq, 8, 0.9), (single, 24, 0.7)]
    twin = torch.Generator(device=DEV).manual_seed(31)
    want = oc.prepare_conditioning(
        [oc.ConditioningItem(m.float(), f, s) for m, f, s in spec], init.clone(), F_, H, W,
        encode=lambda m: oe.vae_encode(sd, cfg, m),
        noise_fn=lambda shape: torch.randn(tuple(shape), generator=twin, device=DEV, dtype=torch.float32).cpu())
    got = pipe.prepare_conditioning([ltxmi.ConditioningItem(m.to(DEV), f, s) for m, f, s in spec],
                                    init.to(DEV).clone(), F_, H, W, vae_per_channel_normalize=True,
                                    generator=torch.Generator(device=DEV).manual_seed(31), sample_posterior=False)
    assert got[3] == want[3] == 18 and got[0].shape == want[0].shape == (1, 48, 128)
    assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[2].cpu(), want[2])
    e = rel(got[0], want[0])
    print(f"prepare_conditioning latents rel L2 {e:.3e}")
    assert e < 1.5e-2                      # bf16 encoder vs fp32 oracle encoder on the conditioned tokens


def test_pipeline_image_to_video_two_steps():
    """i2v loop: conditioning tokens, per-token timestep, image-conditioning noise, masked Euler step
    (pipeline_ltx_video.py:1067-1259) against the oracle's fp32 loop on the same noise draws."""
    import ltxmi
    from oracle import conditioning as oc, dit, sched, vae_encoder as oe
    heads, dh, layers, caption, T = 2, 64, 2, 128, 32
    # causal_temporal_positioning = True as in the 0.9.5+ checkpoints' metadata (the t2v tests run the 2B default, False)
    cfg = dict(dit.default_2b_config(), num_attention_heads=heads, attention_head_dim=dh, num_layers=layers,
               cross_attention_dim=heads * dh, caption_channels=caption, causal_temporal_positioning=True)
    sd32 = {k: v.to(BF).float() for k, v in dit.init_state_dict(cfg, seed=7).items()}
    vcfg, vsd = vae_case("b", with_encoder=True)
    H, W, F_ = 64, 96, 17
    f, h, w = 3, 2, 3
    g = torch.Generator().manual_seed(40)
    lat0 = torch.randn(1, f * h * w, 128, generator=g)
    pos, neg = torch.randn(1, T, caption, generator=g).to(BF), torch.randn(1, T, caption, generator=g).to(BF)
    pmask, nmask = torch.ones(1, T), torch.ones(1, T)
    pmask[:, 20:] = 0
    nmask[:, 5:] = 0
    img, single = [(torch.rand(1, 3, 1, H, W, generator=g) * 2 - 1).to(BF) for _ in range(2)]
    spec = [(img, 0, 1.0), (single, 8, 0.8)]
    gs, stg, rs, skip_blocks, steps, ns = 3.0, 1.0, 0.7, [1], 2, 0.15

    # ---- oracle loop (fp32 = truth; DiT in bf16 = the reference's eager rendering), noise drawn from a twin of the
    # device generator in the product's order
    def oracle(dtype):
        twin = torch.Generator(device=DEV).manual_seed(41)
        draw = lambda shape: torch.randn(tuple(shape), generator=twin, device=DEV, dtype=torch.float32).cpu()  # noqa: E731
        sd = {k: v.to(dtype) for k, v in sd32.items()}
        tsch = sched.set_timesteps(steps, (1, 128, f, h, w))
        lat, pc, mask, n_extra = oc.prepare_conditioning(
            [oc.ConditioningItem(m.float(), fr, s) for m, fr, s in spec], sched.unpatchify(lat0, f, h, w).clone(),
            F_, H, W, encode=lambda m: oe.vae_encode(vsd, vcfg, m), noise_fn=draw,
            causal_fix=cfg["causal_temporal_positioning"])
        frac = pc.to(torch.float32)
        frac[:, 0] = frac[:, 0] * (1.0 / 25.0)
        fc = dit.precompute_freqs_cis(frac, cfg, dtype)
        skip = dit.create_skip_layer_mask(layers, 1, 3, 2, skip_blocks, dtype)
        emb = torch.cat([neg, pos, pos]).to(dtype)
        msk = torch.cat([nmask, pmask, pmask])
        init = lat.clone()
        for t in tsch:
            lat = oc.add_noise_to_image_conditioning_latents(t, init, lat, ns, mask, draw(lat.shape))
            cur_t = oc.per_token_timestep(t, mask, 3)
            npred = dit.transformer3d_forward(sd, cfg, torch.cat([lat.to(dtype)] * 3), fc, emb, cur_t,
                                              encoder_attention_mask=msk, latent_shape=(f, h, w),
                                              skip_layer_mask=skip, skip_layer_strategy=dit.ATTENTION_VALUES)
            v = sched.guidance(npred.float(), 3, gs, stg, rs, True, True, True)
            lat = sched.denoising_step(tsch, lat, v, cur_t[:1], mask, t)
        assert n_extra == h * w
        return sched.unpatchify(lat[:, n_extra:], f, h, w), n_extra

    (truth, n_extra), (eager, _) = oracle(torch.float32), oracle(BF)

    m = build_model(cfg, sd32)
    pipe = ltxmi.LTXVideoPipeline(transformer=m, scheduler=ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1),
                                  vae=build_vae(vcfg, vsd))
    out = pipe(height=H, width=W, num_frames=F_, frame_rate=25.0, prompt_embeds=pos.to(DEV),
               prompt_attention_mask=pmask.to(DEV),
               negative_prompt_embeds=neg.to(DEV), negative_prompt_attention_mask=nmask.to(DEV),
               num_inference_steps=steps, guidance_scale=gs, stg_scale=stg, rescaling_scale=rs,
               skip_block_list=skip_blocks, latents=lat0.to(DEV), output_type="latent", skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, is_video=True, joint_pass=True, latents_dtype=torch.float32,
               conditioning_items=[ltxmi.ConditioningItem(mm.to(DEV), fr, s) for mm, fr, s in spec],
               image_cond_noise_scale=ns, sample_conditioning_posterior=False,
               generator=torch.Generator(device=DEV).manual_seed(41))
    assert out.shape == truth.shape == (1, 128, f, h, w)
    # the conditioning latents come from the bf16 encoder kernels on our side and from the fp32 oracle encoder on both
    # oracle sides (measured alone in test_vae_encode: ~8e-3 vs the eager encoder's ~1.1e-2); that part of our error is
    # not in `eager`, hence the wider additive term here
    e, e_ref = rel(out, truth), rel(eager, truth)
    e0 = rel(out[:, :, 0], truth[:, :, 0])          # the hard-conditioned first frame stays the (noised) encoder output
    print(f"i2v pipeline 2 steps: rel L2 ours {e:.3e} / eager DiT {e_ref:.3e} (first frame {e0:.3e})")
    assert torch.isfinite(out.float()).all()
    assert e <= e_ref + 5 * RTOL and e0 <= 1.4e-2 + RTOL


def test_multiscale_pipeline_two_passes():
    """LTXMultiScalePipeline (pipeline_ltx_video.py:1741-1905): pass 1 with per-step guidance tables at half
    size -> LatentUpsampler -> AdaIN -> pass 2 (re-noised latents, skipped initial steps, joint_pass=False),
    against the oracle's fp32 restatement on the same noise draws."""
    import ltxmi
    from oracle import dit, pipeline_ctl as pc, sched, upsampler as ou
    heads, dh, layers, caption, T = 2, 64, 2, 128, 32
    cfg = dict(dit.default_2b_config(), num_attention_heads=heads, attention_head_dim=dh, num_layers=layers,
               cross_attention_dim=heads * dh, caption_channels=caption)
    sd32 = {k: v.to(BF).float() for k, v in dit.init_state_dict(cfg, seed=7).items()}
    ucfg = dict(in_channels=128, mid_channels=64, num_blocks_per_stage=1, dims=3, spatial_upsample=True,
                temporal_upsample=False)
    usd = {k: v.to(BF).float() for k, v in ou.init_state_dict(ucfg, seed=6).items()}
    g = torch.Generator().manual_seed(50)
    stats = {"per_channel_statistics.std-of-means": 0.5 + torch.rand(128, generator=g),
             "per_channel_statistics.mean-of-means": 0.2 * torch.randn(128, generator=g)}
    pos, neg = torch.randn(1, T, caption, generator=g).to(BF), torch.randn(1, T, caption, generator=g).to(BF)
    pmask, nmask = torch.ones(1, T), torch.ones(1, T)
    pmask[:, 20:] = 0
    nmask[:, 5:] = 0
    first = dict(num_inference_steps=3, guidance_scale=[1, 3], stg_scale=[0, 1], rescaling_scale=[1, 0.7],
                 skip_block_list=[[], [1]], guidance_timesteps=[1.0, 0.95])
    second = dict(num_inference_steps=4, skip_initial_inference_steps=2, guidance_scale=1, stg_scale=0,
                  rescaling_scale=1, skip_block_list=None, guidance_timesteps=None)
    f, h, w = 2, 2, 3                                             # pass-1 latent grid (64 x 96 px, 9 frames)

    twin = torch.Generator(device=DEV).manual_seed(51)
    draw = lambda shape: torch.randn(tuple(shape), generator=twin, device=DEV, dtype=torch.float32).cpu()  # noqa: E731

    def oracle_pass(lat, f, h, w, ts, kw, dtype):
        gs, stg, rs, skips, do_cfg, do_stg, do_rs = pc.guidance_tables(
            [float(x) for x in ts], kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"],
            kw["skip_block_list"], kw["guidance_timesteps"])
        nc = 1 + int(do_cfg) + int(do_stg)
        sd = {k: v.to(dtype) for k, v in sd32.items()}
        emb = torch.cat(([neg] if do_cfg else []) + [pos] + ([pos] if do_stg else [])).to(dtype)
        msk = torch.cat(([nmask] if do_cfg else []) + [pmask] + ([pmask] if do_stg else []))
        pix = sched.latent_to_pixel_coords(sched.get_latent_coords(f, h, w, 1),
                                           causal_fix=cfg.get("causal_temporal_positioning", False)).to(torch.float32)
        pix[:, 0] = pix[:, 0] * (1.0 / 25.0)
        fc = dit.precompute_freqs_cis(pix, cfg, dtype)
        for i, t in enumerate(ts):
            skip = dit.create_skip_layer_mask(layers, 1, nc, nc - 1, skips[i], dtype) if (do_stg and skips) else None
            npred = dit.transformer3d_forward(sd, cfg, torch.cat([lat.to(dtype)] * nc), fc, emb, t.expand(nc).unsqueeze(-1),
                                              encoder_attention_mask=msk, latent_shape=(f, h, w),
                                              skip_layer_mask=skip, skip_layer_strategy=dit.ATTENTION_VALUES)
            v = sched.guidance(npred.float(), nc, gs[i], stg[i], rs[i], do_cfg, do_stg, do_rs)
            lat = sched.denoising_step(ts, lat, v, t.expand(1).unsqueeze(-1), None, t)
        return sched.unpatchify(lat, f, h, w)

    shape2 = (1, 128, f, 2 * h, 2 * w)
    ts1 = pc.retrieve_timesteps(3, (1, 128, f, h, w))
    ts2 = pc.retrieve_timesteps(4, shape2, skip_initial_inference_steps=2)
    n1, n2 = draw((1, f * h * w, 128)), draw((1, f * 4 * h * w, 128))

    def oracle_two_passes(dtype):                       # the upsampler bridge stays fp32 on both oracle sides
        lat1 = oracle_pass(n1, f, h, w, ts1, first, dtype)
        up = ou.adain_filter_latent(ou.upsample_latents(usd, ucfg, lat1, stats), lat1)
        start = pc.prepare_latents(up, float(ts2[0]), n2, shape2)
        return oracle_pass(sched.patchify(start)[0], f, 2 * h, 2 * w, ts2, second, dtype)

    truth, eager = oracle_two_passes(torch.float32), oracle_two_passes(BF)

    m = build_model(cfg, sd32)
    ups = ltxmi.LatentUpsampler.from_config(ucfg)
    ups.load_state_dict(usd)
    ups = ups.to(device=DEV, dtype=BF).eval()
    vae = types.SimpleNamespace(std_of_means=stats["per_channel_statistics.std-of-means"].to(DEV),
                                mean_of_means=stats["per_channel_statistics.mean-of-means"].to(DEV))
    vp = ltxmi.LTXVideoPipeline(transformer=m, scheduler=ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1), vae=vae)
    ms = ltxmi.LTXMultiScalePipeline(vp, ups)
    out = ms(0.5, first, second, height=128, width=192, num_frames=9, frame_rate=25.0, prompt_embeds=pos.to(DEV),
             prompt_attention_mask=pmask.to(DEV), negative_prompt_embeds=neg.to(DEV),
             negative_prompt_attention_mask=nmask.to(DEV), output_type="latent",
             skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, is_video=True, latents_dtype=torch.float32,
             generator=torch.Generator(device=DEV).manual_seed(51))
    assert out.shape == truth.shape == shape2
    torch.testing.assert_close(torch.tensor(vp.scheduler.host_timesteps), ts2, rtol=1e-6, atol=1e-7)
    # our bf16 LatentUpsampler kernels sit between the passes (alone: 7.7e-3 vs the eager upsampler's 1.03e-2,
    # test_latent_upsampler_and_bridge); both oracle sides run that bridge in fp32, hence the wider additive term
    e, e_ref = rel(out, truth), rel(eager, truth)
    print(f"multi-scale 2 passes: rel L2 ours {e:.3e} / eager DiT {e_ref:.3e}")
    assert torch.isfinite(out.float()).all() and e <= e_ref + 5 * RTOL
    # the exact eliminations (rows whose guidance scale is zero at a step; the STG row before its first
    # skipped block) do not change a single bit
    plain = ms(0.5, first, second, height=128, width=192, num_frames=9, frame_rate=25.0, prompt_embeds=pos.to(DEV),
               prompt_attention_mask=pmask.to(DEV), negative_prompt_embeds=neg.to(DEV),
               negative_prompt_attention_mask=nmask.to(DEV), output_type="latent",
               skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, is_video=True, latents_dtype=torch.float32,
               generator=torch.Generator(device=DEV).manual_seed(51), stg_row_dedup=False,
               dead_row_elimination=False)
    assert torch.equal(plain, out)


def test_ltxv_generate_call_replayed_through_the_product(golden):
    """The drop-in boundary, end to end: the keyword arguments of ``LTXV.generate``'s pipeline call (ltxv.py:420-445 -- the
    YAML dict spread into the call, string prompts, ``output_type="pt"``, ``VAE_tile_size``, ``device``, ``callback`` ...) go
    UNCHANGED into ``ltxmi.LTXMultiScalePipeline`` and the result is compared with what the reference's own
    ``LTXMultiScalePipeline.__call__`` produced for them (golden G15, fp32 on the CPU), within what the reference's bf16
    eager path manages (the oracle's two-pass restatement, itself pinned to G15 by tests/test_oracle_golden.py, run in
    bf16).  The noise draws are the recorded ones (the reference drew them on the CPU); the T5 pair is the same test
    double that drove the reference."""
    import ltxmi
    from unittest import mock
    from oracle import pipeline_ctl as pc
    from test_oracle_golden import g15_case
    t, meta, sd32, vsd, usd, tok, enc = g15_case(golden)
    call, cfgp = dict(meta["call"]), meta["pipeline_config"]
    pos, neg = t["prompt_embeds"], t["negative_prompt_embeds"]
    truth = t["images"]
    eager, _, _ = pc.multiscale_call(
        sd32, meta["dit_cfg"], vsd, meta["vae_cfg"], usd, meta["upsampler_cfg"], pos, neg, t["prompt_attention_mask"],
        t["negative_prompt_attention_mask"], call["height"], call["width"], call["num_frames"], call["frame_rate"],
        cfgp["downscale_factor"], cfgp["first_pass"], cfgp["second_pass"], call["num_inference_steps1"],
        call["num_inference_steps2"], t["noise.0"], t["noise.1"], t["decode_noise"], cfgp["decode_timestep"],
        cfgp["decode_noise_scale"], dtype=BF, vae_dtype=BF, stats=vsd)

    m = build_model(meta["dit_cfg"], sd32)
    vae = build_vae(dict(meta["vae_cfg"], build_encoder=False), vsd)
    ups = ltxmi.LatentUpsampler.from_config(meta["upsampler_cfg"])
    ups.load_state_dict(usd)
    ups = ups.to(device=DEV, dtype=BF).eval()
    # the constructor call of ltxv.py:219-233, keyword for keyword
    pipe = ltxmi.LTXVideoPipeline(
        transformer=m, patchifier=ltxmi.SymmetricPatchifier(patch_size=1), text_encoder=enc.to(DEV), tokenizer=tok,
        scheduler=ltxmi.RectifiedFlowScheduler(sampler="Uniform", shifting="SD3", base_resolution=None, target_shift_terminal=0.1),
        vae=vae, prompt_enhancer_image_caption_model=None, prompt_enhancer_image_caption_processor=None,
        prompt_enhancer_llm_model=None, prompt_enhancer_llm_tokenizer=None, allowed_inference_steps=None)
    ms = ltxmi.LTXMultiScalePipeline(pipe, ups)
    seen = {}
    up = ms._upsample_latents

    def spy(upsampler, latents):
        out = up(upsampler, latents)
        seen["pass1"], seen["upsampled"] = latents.float().cpu(), out.float().cpu()
        return out

    ms._upsample_latents = spy
    trace = []

    def callback(i, preview, start, **kw):
        trace.append([int(i), None if preview is None else list(preview.shape), bool(start), int(kw.get("pass_no", 0)),
                      kw.get("override_num_inference_steps")])

    draws = [t["noise.0"], t["noise.1"]]
    plain_randn = torch.randn

    def recorded_randn(shape, *a, generator=None, device=None, dtype=None, **k):
        n = draws.pop(0)
        assert tuple(shape) == tuple(n.shape)
        return n.to(device=device, dtype=dtype)

    call["device"] = DEV                                      # (the reference ran with device="cpu")
    with mock.patch.object(torch, "randn", recorded_randn), \
            mock.patch.object(torch, "randn_like", lambda x, **k: t["decode_noise"].to(device=x.device, dtype=x.dtype)):
        images = ms(**cfgp, ltxv_model=types.SimpleNamespace(_interrupt=False),
                    skip_layer_strategy=ltxmi.SkipLayerStrategy[meta["skip_layer_strategy"]],
                    generator=torch.Generator(device=DEV).manual_seed(155), callback=callback, **call)
    assert torch.randn is plain_randn and not draws
    assert trace == meta["callback_trace"]
    assert images.shape == truth.shape == tuple(meta["images_shape"]) and images.dtype == BF
    e1 = rel(seen["pass1"], t["pass1_latents"])
    print(f"G15 replay: pass-1 latents rel L2 {e1:.3e}, upsampled {rel(seen['upsampled'], t['upsampled']):.3e}")
    assert_parity(images, truth, eager, "ltxv.py's pipeline call through ltxmi.LTXMultiScalePipeline vs the reference's own run")
    # an interrupt raised before the first pass comes back as None, as in the reference (:1854-1855)
    call2 = dict(call)
    assert ms(**cfgp, ltxv_model=types.SimpleNamespace(_interrupt=True), callback=None,
              skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, generator=None, **call2) is None


def test_ulysses_processor_world1_matches_default_processor():
    """The sequence-parallel code path (UlyssesAttnProcessor + usp_dit_forward) on the GPU with a
    1-rank RCCL group: the all-to-alls are identities, so the result must equal the default
    processor's bit for bit (same kernels, same order)."""
    import os
    import torch.distributed as dist
    from ltxmi import distributed as sp
    grid, B, T = (2, 4, 8), 3, 24
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 2, grid, B, T, seed=11)
    m = build_model(cfg, sd32)
    fc = m.precompute_freqs_cis(frac.to(DEV))
    kw = dict(encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV), timestep=ts.to(DEV),
              latent_shape=grid, ltxv_model=_Holder())
    ref = m(x.to(DEV), freqs_cis=fc, return_dict=False, **kw)[0]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        sp.enable_sequence_parallel(m)
        out = sp.usp_dit_forward(m, x.to(DEV), fc, **kw)[0]
        torch.cuda.synchronize()
        # the identity short-cut off: dist.all_to_all_single (RCCL) really runs on the zero-copy send / receive buffers, the
        # attention kernel reads the receive buffer through its strides, to_out takes the return buffer as a K-blocked
        # operand -- on this one GPU, with a one-rank group (a copy by RCCL): still the default processor's bits
        sp.enable_sequence_parallel(m, exchange_at_world_1=True)
        assert all(b.attn1.processor.exchange_at_world_1 for b in m.transformer_blocks)
        calls = []
        a2a = dist.all_to_all_single

        def counted(recv, send, **k):
            calls.append((tuple(send.shape), send.is_cuda))
            return a2a(recv, send, **k)

        dist.all_to_all_single = counted
        try:
            forced = sp.usp_dit_forward(m, x.to(DEV), fc, **kw)[0]
            torch.cuda.synchronize()
        finally:
            dist.all_to_all_single = a2a
        assert len(calls) == 2 * len(m.transformer_blocks) and all(c[1] for c in calls)
        # and as the pipeline sees it: model.forward rebound (the reference's MethodType pattern)
        sp.enable_sequence_parallel(m, bind_forward=True, exchange_at_world_1=True)
        bound = m(x.to(DEV), freqs_cis=fc, return_dict=False, **kw)[0]
        torch.cuda.synchronize()
        sp.disable_sequence_parallel(m)
        again = m(x.to(DEV), freqs_cis=fc, return_dict=False, **kw)[0]
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.equal(out, ref)
    assert torch.equal(forced, ref)
    assert torch.equal(bound, ref) and torch.equal(again, ref)


def test_microbatched_block_loop_as_the_first_forward_of_a_fresh_model():
    """the micro-batched block loop (two row slices, a side stream each -- the overlap mode of sequence
    parallelism, reachable at world size 1 through ``_microbatches``) as the VERY FIRST forward of a fresh model: the packed
    projection weights are built on the main stream before the side streams fork, so slice 1 can never read a pack that
    stream 0 is still writing.  Compared bit for bit with a second fresh model run without micro-batches; again after the
    packs were invalidated; again with step-invariant caching off (the stacked text K/V then serve every slice); and with
    ``stg_alias_blocks`` (which takes the plain loop)."""
    import ltxmi
    grid, B, T = (2, 4, 8), 3, 24
    cfg, sd32, x, enc, mask, ts, frac = dit_case(2, 64, 3, grid, B, T, seed=12)
    x[2], enc[2], mask[2], ts[2] = x[1], enc[1], mask[1], ts[1]            # the STG row repeats the text row's inputs
    slices = [slice(0, 1), slice(1, 3)]

    def run(model, **extra):
        fc = model.precompute_freqs_cis(frac.to(DEV))
        out = model(x.to(DEV).clone(), freqs_cis=fc, encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV),
                    timestep=ts.to(DEV), latent_shape=grid, ltxv_model=_Holder(), return_dict=False, **extra)[0]
        torch.cuda.synchronize()
        return out

    fresh = build_model(cfg, sd32)
    assert all(not b.attn1._packs and not b.attn2._packs for b in fresh.transformer_blocks)
    first = run(fresh, _microbatches=slices)                               # nothing was ever packed before this call
    plain = run(build_model(cfg, sd32))
    assert torch.equal(first, plain)
    for b in fresh.transformer_blocks:
        b.attn1.invalidate_packed()
        b.attn2.invalidate_packed()
    assert torch.equal(run(fresh, _microbatches=slices), plain)
    ltxmi.set_step_invariant_caching(False)
    try:
        assert torch.equal(run(build_model(cfg, sd32), _microbatches=slices), plain)
        assert torch.equal(run(fresh, _microbatches=slices, stg_alias_blocks=2), plain)
    finally:
        ltxmi.set_step_invariant_caching(True)
    assert torch.equal(run(fresh, _microbatches=slices, stg_alias_blocks=2), plain)


# ------------------------------------------------------------------------------- Ulysses at world size 2, real kernels
def _host_staged_collectives():
    """gloo moves CPU tensors only: for device tensors the three collectives of ltxmi.distributed are staged through
    the host IN THE TEST WORKERS (the product code is untouched; on a multi-GPU node the same calls go to RCCL)."""
    import torch.distributed as dist
    a2a, gather, reduce = dist.all_to_all_single, dist.all_gather, dist.all_reduce

    def all_to_all_single(recv, send, group=None, **kw):
        if not send.is_cuda:
            return a2a(recv, send, group=group, **kw)
        r, s = torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
        a2a(r, s, group=group, **kw)
        recv.copy_(r)

    def all_gather(parts, x, group=None, **kw):
        if not x.is_cuda:
            return gather(parts, x, group=group, **kw)
        host = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        gather(host, x.cpu(), group=group, **kw)
        for p, h in zip(parts, host):
            p.copy_(h)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, **kw):
        if not t.is_cuda:
            return reduce(t, op=op, group=group, **kw)
        h = t.cpu()
        reduce(h, op=op, group=group, **kw)
        t.copy_(h)

    gather_flat = dist.all_gather_into_tensor

    def all_gather_into_tensor(out, x, group=None, **kw):
        if not x.is_cuda:
            return gather_flat(out, x, group=group, **kw)
        h = torch.empty(out.shape, dtype=out.dtype)
        gather_flat(h, x.cpu(), group=group, **kw)
        out.copy_(h)

    dist.all_to_all_single, dist.all_gather, dist.all_reduce = all_to_all_single, all_gather, all_reduce
    dist.all_gather_into_tensor = all_gather_into_tensor


def _ulysses_world2_worker(rank, world, port, per_token, q):
    import os
    import sys
    import traceback
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ltx-video-gpupoor_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)                                  # both ranks share the one GPU of the box
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _host_staged_collectives()
        import ltxmi
        from ltxmi import distributed as sp
        if per_token == "p8shape":
            # what ONE rank sees at P = 8 on the 2B model, reproduced at P = 2: 8 heads of 64 -> H/P = 4 local heads,
            # D/P = 256-wide exchange blocks, 3 x 4 x 16 = 192 workgroups of 256 query rows -> the pipelined attention
            # kernel writing the segmented output
            grid, B, T, heads, per_token = (4, 32, 32), 3, 32, 8, False
            from ltxmi import ops
            assert ops.attention_kernel_id(B, heads // world, 4096, 4096, 64) == ops.attention_kernel_id(3, 32, 4992, 4992, 64)
        else:
            grid, B, T, heads = (4, 16, 16), 3, 32, 4                 # N = 1024 tokens, 512 per rank, 2 frames per rank
        cfg, sd32, x, enc, mask, ts, frac = dit_case(heads, 64, 2, grid, B, T, seed=31, per_token=per_token)
        m = build_model(cfg, sd32)
        fc = m.precompute_freqs_cis(frac.to(DEV))
        kw = dict(encoder_hidden_states=enc.to(DEV), encoder_attention_mask=mask.to(DEV), timestep=ts.to(DEV),
                  skip_layer_mask=m.create_skip_layer_mask(1, 3, 2, [1]),
                  skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=grid)
        with torch.no_grad():
            ref = m(x.to(DEV).clone(), freqs_cis=fc, return_dict=False, **kw)[0]      # one rank, default processor
            sp.enable_sequence_parallel(m, overlap=False)
            plain = sp.usp_dit_forward(m, x.to(DEV).clone(), fc, **kw)[0]             # tokens sharded over 2 ranks
            sp.enable_sequence_parallel(m)                                            # + rows as two micro-batches on two streams
            out = sp.usp_dit_forward(m, x.to(DEV).clone(), fc, **kw)[0]
        torch.cuda.synchronize()
        err = float((out.float() - ref.float()).norm() / ref.float().norm())
        assert out.shape == ref.shape and torch.isfinite(out.float()).all()
        # the overlap mode (micro-batches of rows on side streams) changes no bit when both modes take the same attention
        # kernel; at the small shape the row split moves attention below the pipelined kernel's threshold: close instead
        e_mb = float((out.float() - plain.float()).norm() / plain.float().norm())
        assert e_mb < 2e-3, e_mb
        # same kernels on the same rows; only the GEMM tile positions and the heads-per-launch of attention differ
        assert err < 2e-3, err
        q.put((rank, "ok", err))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.parametrize("per_token", [False, True, "p8shape"])
def test_ulysses_world2_real_kernels_on_one_gpu(per_token):
    """usp_dit_forward + UlyssesAttnProcessor at WORLD SIZE 2 with the real kernels: two processes share the box's one
    GPU, the all-to-alls travel over gloo (host-staged in the workers).  Everything device-side is what runs on a
    multi-GPU node -- the destination-major pack kernel with P = 2, the token-major strided attention input, the
    segmented attention output, the K-blocked to_out GEMM, the sliced RoPE tables, per-token timesteps -- only the
    transport differs.  Against the single-rank forward of the same model."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ulysses_world2_worker, args=(r, 2, port, per_token, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, err in results:
        assert status == "ok", f"rank {rank}:\n{status}"


def _tile_parallel_decode_worker(rank, world, port, q):
    import os
    import sys
    import traceback
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ltx-video-gpupoor_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gather = dist.all_gather_into_tensor

        def all_gather_into_tensor(out, x, group=None, **kw):      # gloo moves host tensors: staged in the test worker
            if not x.is_cuda:
                return gather(out, x, group=group, **kw)
            h = torch.empty(out.shape, dtype=out.dtype)
            gather(h, x.cpu(), group=group, **kw)
            out.copy_(h)
        dist.all_gather_into_tensor = all_gather_into_tensor
        import ltxmi
        from ltxmi import distributed as sp
        cfg, sd = vae_case("b", base=64)
        v = build_vae(cfg, sd)
        v.enable_z_tiling(4)
        z = torch.randn(1, 128, 10, 2, 3, generator=torch.Generator().manual_seed(14)).to(BF).to(DEV)   # 4 z-tiles
        ts = torch.tensor([0.05], device=DEV)
        with torch.no_grad():
            ref = ltxmi.vae_decode(z, v, True, vae_per_channel_normalize=True, timestep=ts)         # every tile on this rank
            trace = []
            out = sp.tile_parallel_vae_decode(z, v, True, vae_per_channel_normalize=True, timestep=ts, _trace=trace)
        torch.cuda.synchronize()
        assert out.shape == ref.shape == (1, 3, 73, 64, 96) and out.dtype == torch.float16
        assert torch.equal(out, ref)
        # concurrency, structurally: this rank's tiles (n = rank mod 2) are ALL decoded before the one collective is
        # issued -- no collective sits between two local decodes, so the ranks' decodes overlap in time
        assert trace == [("decode", n) for n in range(rank, 4, world)] + [("collective", "all_gather_into_tensor")], trace
        q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


def test_tile_parallel_vae_decode_world2_on_one_gpu():
    """SURVEY 8e, VAE row: the z-tiles of a tiled decode spread over the ranks (rank r decodes tiles r, r + P, ... back to
    back, ONE all-gather, blends on every rank) -- world size 2, both ranks on the box's one GPU, gloo transport -- gives
    the single-rank tiled decode bit for bit on every rank, with no collective issued before the last local decode."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tile_parallel_decode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status in results:
        assert status == "ok", f"rank {rank}:\n{status}"
