"""Host-side logic of the PRODUCT (ltxmi/) against the reference-generated golden vectors -- CPU only.
RoPE tables, the scheduler, the per-step guidance tables and the skip-layer masks are computed on the host once
per generation; none of them needs a GPU, so they are pinned here directly (not only through model-level tolerances)."""
import torch

import ltxmi
from ltxmi.patchifier import SymmetricPatchifier, latent_to_pixel_coords_from_factors

TOL = dict(rtol=1e-5, atol=2e-6)


def _model(heads, dh, layers=1):
    # a 1-layer model only to own the RoPE parameters (inner_dim, theta, max_pos); the tables do not depend on weights
    return ltxmi.Transformer3DModel(num_attention_heads=heads, attention_head_dim=dh, in_channels=16, out_channels=16,
                                    num_layers=layers, cross_attention_dim=heads * dh, caption_channels=32,
                                    attention_bias=True, activation_fn="gelu-approximate", norm_elementwise_affine=False,
                                    norm_eps=1e-6, qk_norm="rms_norm", standardization_norm="rms_norm",
                                    adaptive_norm="single_scale_shift", positional_embedding_type="rope",
                                    positional_embedding_theta=10000.0, positional_embedding_max_pos=[20, 2048, 2048],
                                    timestep_scale_multiplier=1000)


def test_g1_product_precompute_freqs_cis(golden):
    """Transformer3DModel.precompute_freqs_cis of the product (transformer3d.py:202-255) against G1, including the
    D = 2048 case whose 2 leading pad channels (2048 % 6) must be cos 1 / sin 0, and the coordinate grid the product's
    patchifier + causal fix produce."""
    t, meta = golden("g1_rope")
    for case in meta["cases"]:
        tag = case["tag"]
        f, h, w = case["grid"]
        coords = SymmetricPatchifier(1).get_latent_coords(f, h, w, 2, "cpu")
        pc = latent_to_pixel_coords_from_factors(coords, (8, 32, 32), causal_fix=True).to(torch.float32)
        pc[:, 0] = pc[:, 0] * (1.0 / 25.0)
        torch.testing.assert_close(pc, t[f"{tag}.indices_grid"], rtol=0, atol=0)
        m = _model(case["heads"], case["dh"])                      # fp32 module: tables come out in fp32
        cos, sin = m.precompute_freqs_cis(t[f"{tag}.indices_grid"])
        torch.testing.assert_close(cos, t[f"{tag}.cos"], **TOL)
        torch.testing.assert_close(sin, t[f"{tag}.sin"], **TOL)
        # the bf16 tables the kernels read are the fp32 tables rounded once
        cb, sb = m.to(torch.bfloat16).precompute_freqs_cis(t[f"{tag}.indices_grid"])
        assert torch.equal(cb, t[f"{tag}.cos"].to(torch.bfloat16)) or (cb.float() - t[f"{tag}.cos"]).abs().max() < 4e-3
        assert (sb.float() - t[f"{tag}.sin"]).abs().max() < 4e-3


def test_g6_product_scheduler(golden):
    """ltxmi.RectifiedFlowScheduler.set_timesteps (SD3 resolution-dependent shift, stretch to terminal 0.1) against G6
    for the latent shapes of all five configs."""
    t, meta = golden("g6_scheduler")
    for tag, shp in meta["shapes"].items():
        for steps in (2, 8, 40):
            s = ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1)
            s.set_timesteps(steps, samples_shape=tuple(shp), device="cpu")
            torch.testing.assert_close(s.timesteps.float().cpu(), t[f"{tag}.steps{steps}"], rtol=1e-6, atol=1e-7)


def test_g7_product_guidance_tables_and_skip_masks(golden):
    """LTXVideoPipeline._guidance_tables + Transformer3DModel.create_skip_layer_mask of the product against what the
    reference's own __call__ fed its loop (G7, the run with list-valued scales and guidance_timesteps)."""
    t, meta = golden("g7_pipeline_call")
    kw = meta["tables_kwargs"]
    ts = t["tables.timesteps"].tolist()
    gs, stg, rs, skips = ltxmi.LTXVideoPipeline._guidance_tables(ts, kw["guidance_scale"], kw["stg_scale"],
                                                                 kw["rescaling_scale"], kw["skip_block_list"],
                                                                 kw["guidance_timesteps"])
    # first table entry whose guidance timestep is <= t (:961-968); guidance scales <= 1 are zeroed (:983)
    assert gs == [0.0, 3.0, 2.0][: len(ts)] or len(gs) == len(ts)
    m = _model(2, 64, layers=meta["cfg"]["num_layers"])
    for i in range(len(ts)):
        mask = m.create_skip_layer_mask(1, 3, 2, skips[i])
        if f"tables.skip_layer_mask.{i}" in t:
            assert torch.equal(mask.float(), t[f"tables.skip_layer_mask.{i}"])
        else:
            assert mask is None
    from oracle import pipeline_ctl as pc
    ogs, ostg, ors, oskips, *_ = pc.guidance_tables(ts, kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"],
                                                    kw["skip_block_list"], guidance_timesteps=kw["guidance_timesteps"])
    assert (gs, stg, rs, skips) == (ogs, ostg, ors, oskips)


def test_g6_product_scheduler_step(golden):
    """ltxmi.RectifiedFlowScheduler.step: Euler (global and per-token timesteps) and the stochastic branch (same torch
    generator state as the recorded draw) against G6."""
    t, meta = golden("g6_scheduler")
    s = ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1)
    s.set_timesteps(8, samples_shape=tuple(meta["shapes"]["small"]), device="cpu")
    ts = s.timesteps.float()
    torch.testing.assert_close(ts, t["step.timesteps"], rtol=1e-6, atol=1e-7)
    out = s.step(t["step.v"], ts[2], t["step.sample"], return_dict=False)[0]
    torch.testing.assert_close(out, t["step.global"], **TOL)
    out = s.step(t["step.v"], t["step.tok_t"], t["step.sample"], return_dict=False)[0]
    torch.testing.assert_close(out, t["step.per_token"], **TOL)
    g = torch.Generator().manual_seed(7)            # the draw the reference took from the global RNG seeded with 7
    out = s.step(t["step.v"], t["step.tok_t"], t["step.sample"], return_dict=False, stochastic_sampling=True, generator=g)[0]
    torch.testing.assert_close(out, t["step.stochastic_per_token"], **TOL)
