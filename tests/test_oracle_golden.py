"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference's
own code (oracle/gen/make_golden.py, run in the build container).  CPU only."""
import torch

from oracle import dit, vae, sched

TOL = dict(rtol=1e-5, atol=2e-6)   # fp32 vs fp32, different op order in a few places


def sub(t, prefix):
    return {k[len(prefix):]: v for k, v in t.items() if k.startswith(prefix)}


def test_g1_g2_rope(golden):
    t, meta = golden("g1_rope")
    for case in meta["cases"]:
        tag = case["tag"]
        cfg = dict(dit.default_2b_config(), num_attention_heads=case["heads"],
                   attention_head_dim=case["dh"])
        # indices grid must match the patchifier + causal-fix restatement
        f, h, w = case["grid"]
        pc = sched.fractional_coords(f, h, w, 2, 25.0)
        torch.testing.assert_close(pc, t[f"{tag}.indices_grid"], rtol=0, atol=0)
        cos, sin = dit.precompute_freqs_cis(t[f"{tag}.indices_grid"], cfg, torch.float32)
        torch.testing.assert_close(cos, t[f"{tag}.cos"], **TOL)
        torch.testing.assert_close(sin, t[f"{tag}.sin"], **TOL)
        rot = dit.apply_rotary_emb(t[f"{tag}.x"], (t[f"{tag}.cos"], t[f"{tag}.sin"]))
        torch.testing.assert_close(rot, t[f"{tag}.rot"], **TOL)
        if cos.shape[-1] % 6:
            pad = cos.shape[-1] % 6
            assert (cos[..., :pad] == 1).all() and (sin[..., :pad] == 0).all()


def test_g3_attention_processor(golden):
    t, meta = golden("g3_attention")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    fc = (t["cos"], t["sin"])
    p = "transformer_blocks.0."
    out = dit.attention_processor(sd, p + "attn1.", cfg, t["hs"], freqs_cis=fc)
    torch.testing.assert_close(out, t["self"], **TOL)
    out = dit.attention_processor(sd, p + "attn2.", cfg, t["hs"], freqs_cis=fc,
                                  encoder_hidden_states=t["ctx"], attention_mask=t["bias"])
    torch.testing.assert_close(out, t["cross"], **TOL)
    out = dit.attention_processor(sd, p + "attn1.", cfg, t["hs"], freqs_cis=fc, skip_layer_mask=t["slm"],
                                  skip_layer_strategy=dit.ATTENTION_VALUES)
    torch.testing.assert_close(out, t["self_stg_values"], **TOL)
    out = dit.attention_processor(sd, p + "attn1.", cfg, t["hs"], freqs_cis=fc, skip_layer_mask=t["slm"],
                                  skip_layer_strategy=dit.ATTENTION_SKIP)
    torch.testing.assert_close(out, t["self_stg_skip"], **TOL)
    out = dit.attention_processor(sd, p + "attn1.", cfg, t["hs"][:1], freqs_cis=fc,
                                  skip_layer_mask=torch.tensor([0.0]),
                                  skip_layer_strategy=dit.ATTENTION_VALUES)
    torch.testing.assert_close(out, t["self_stg_values_b1"], **TOL)


def test_g4_block(golden):
    t, meta = golden("g4_block")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    fc = (t["cos"], t["sin"])
    p = "transformer_blocks.0."
    kw = dict(freqs_cis=fc, encoder_hidden_states=t["ctx"], encoder_attention_mask=t["bias"])
    slm = torch.tensor([1.0, 1.0, 0.0])
    out = dit.transformer_block(sd, p, cfg, t["hs"], timestep=t["temb_b"], **kw)
    torch.testing.assert_close(out, t["out_b"], **TOL)
    out = dit.transformer_block(sd, p, cfg, t["hs"], timestep=t["temb_f"], **kw)
    torch.testing.assert_close(out, t["out_f"], **TOL)
    out = dit.transformer_block(sd, p, cfg, t["hs"], timestep=t["temb_b"], skip_layer_mask=slm,
                                skip_layer_strategy=dit.ATTENTION_VALUES, **kw)
    torch.testing.assert_close(out, t["out_b_stg"], **TOL)
    out = dit.transformer_block(sd, p, cfg, t["hs"], timestep=t["temb_b"], skip_layer_mask=slm,
                                skip_layer_strategy=dit.TRANSFORMER_BLOCK, **kw)
    torch.testing.assert_close(out, t["out_b_tb"], **TOL)


def test_g5_transformer(golden):
    t, meta = golden("g5_transformer")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    f, h, w = meta["grid"]
    fc = dit.precompute_freqs_cis(t["indices_grid"], cfg, torch.float32)
    kw = dict(encoder_attention_mask=t["mask"], latent_shape=(f, h, w))
    out = dit.transformer3d_forward(sd, cfg, t["x"], fc, t["enc"], t["ts"], **kw)
    torch.testing.assert_close(out, t["out"], **TOL)
    skip = dit.create_skip_layer_mask(cfg["num_layers"], 1, 3, 2, meta["skip_blocks"], torch.float32)
    torch.testing.assert_close(skip, t["skip_layer_mask"], rtol=0, atol=0)
    out = dit.transformer3d_forward(sd, cfg, t["x"], fc, t["enc"], t["ts"], skip_layer_mask=skip,
                                    skip_layer_strategy=dit.ATTENTION_VALUES, **kw)
    torch.testing.assert_close(out, t["out_stg"], **TOL)
    out = dit.transformer3d_forward(sd, cfg, t["x"], fc, t["enc"], t["ts_tok"], **kw)
    torch.testing.assert_close(out, t["out_tok"], **TOL)


def test_g5_transformer_bf16_twin(golden):
    """Run in bf16 the oracle has the reference's eager rounding points: it must
    reproduce the reference's bf16 output to within a couple of bf16 ulps."""
    t, meta = golden("g5_transformer")
    cfg = meta["cfg"]
    bf = torch.bfloat16
    sd = {k: v.to(bf) for k, v in sub(t, "sd.").items()}
    f, h, w = meta["grid"]
    fc = dit.precompute_freqs_cis(t["indices_grid"], cfg, bf)
    torch.testing.assert_close(fc[0], t["bf16.cos"], rtol=0, atol=0)
    out = dit.transformer3d_forward(sd, cfg, t["x"].to(bf), fc, t["enc"].to(bf), t["ts"],
                                    encoder_attention_mask=t["mask"], latent_shape=(f, h, w))
    ref = t["bf16.out"].float()
    err = (out.float() - ref).norm() / ref.norm()
    assert err < 1e-2, err
    # both are bf16 renderings of the same fp32 truth
    truth = t["out"]
    e_ref = (ref - truth).norm() / truth.norm()
    e_orc = (out.float() - truth).norm() / truth.norm()
    assert e_orc < 2 * e_ref + 1e-3, (e_orc, e_ref)


def test_g6_scheduler(golden):
    t, meta = golden("g6_scheduler")
    for tag, shp in meta["shapes"].items():
        for steps in (2, 8, 40):
            ts = sched.set_timesteps(steps, tuple(shp))
            torch.testing.assert_close(ts, t[f"{tag}.steps{steps}"], rtol=1e-6, atol=1e-7)
    tsched = t["step.timesteps"]
    out = sched.scheduler_step(tsched, t["step.v"], tsched[2], t["step.sample"])
    torch.testing.assert_close(out, t["step.global"], **TOL)
    out = sched.scheduler_step(tsched, t["step.v"], t["step.tok_t"], t["step.sample"])
    torch.testing.assert_close(out, t["step.per_token"], **TOL)


def test_g8_g9_conv_blocks(golden):
    t, _ = golden("g8_conv_blocks")
    x = t["x"]
    for mode in ("zeros", "replicate"):
        sd = sub(t, f"conv.{mode}.")
        torch.testing.assert_close(vae.causal_conv3d(x, sd, "", True, mode), t[f"conv.{mode}.causal"], **TOL)
        torch.testing.assert_close(vae.causal_conv3d(x, sd, "", False, mode), t[f"conv.{mode}.noncausal"], **TOL)
    torch.testing.assert_close(vae.pixel_norm(x), t["pixel_norm"], **TOL)
    torch.testing.assert_close(vae.patchify(t["patch.x"], 4, 1), t["patch.patchified"], rtol=0, atol=0)
    torch.testing.assert_close(vae.unpatchify(t["patch.patchified"], 4, 1), t["patch.x"], rtol=0, atol=0)
    torch.testing.assert_close(t["patch.roundtrip"], t["patch.x"], rtol=0, atol=0)
    out = vae.resnet_block(t["res.x"], sub(t, "res.sd."), "", False, "replicate", t["res.temb"])
    torch.testing.assert_close(out, t["res.out"], **TOL)
    out = vae.resnet_block(t["res.x"], sub(t, "resxy.sd."), "", False, "zeros", None)
    torch.testing.assert_close(out, t["resxy.out"], **TOL)
    blk = dict(stride=(2, 2, 2), residual=True, reduction=2)
    out = vae.depth_to_space_upsample(t["res.x"], sub(t, "up.sd."), "", blk, False, "replicate")
    torch.testing.assert_close(out, t["up.out"], **TOL)
    blk = dict(stride=(2, 2, 2), residual=False, reduction=1)
    out = vae.depth_to_space_upsample(t["res.x"], sub(t, "up2.sd."), "", blk, False, "zeros")
    torch.testing.assert_close(out, t["up2.out"], **TOL)


def _decoder_case(golden, tag):
    t, meta = golden(f"g10_decoder_{tag}")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    sd["per_channel_statistics.std-of-means"] = t["per_channel_statistics.std-of-means"]
    sd["per_channel_statistics.mean-of-means"] = t["per_channel_statistics.mean-of-means"]
    ts = t.get("timestep")
    tol = dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(vae.decode(sd, cfg, t["z"], ts), t["decode"], **tol)
    torch.testing.assert_close(vae.vae_decode(sd, cfg, t["z"], ts), t["vae_decode"], **tol)
    out = vae.decode(sd, cfg, t["z_ztile"], ts, use_z_tiling=True, z_sample_size=4)
    assert out.dtype == torch.float16 and out.shape == t["decode_ztile"].shape
    torch.testing.assert_close(out.float(), t["decode_ztile"].float(), rtol=2e-3, atol=2e-3)
    out = vae.decode(sd, cfg, t["z_hwtile"], ts, use_hw_tiling=True, tile_sample_min_size=64)
    torch.testing.assert_close(out, t["decode_hwtile"], **tol)


def test_g10_decoder_plan_a(golden):
    _decoder_case(golden, "a")


def test_g10_decoder_plan_b_timestep_conditioned(golden):
    _decoder_case(golden, "b")


def test_reference_invariants():
    """The reference's own in-module checks, re-expressed:
    unpatchify(patchify(x)) == x (causal_video_autoencoder.py:1341-1347)."""
    x = torch.randn(2, 3, 8, 64, 64)
    assert torch.equal(vae.unpatchify(vae.patchify(x, 4, 4), 4, 4), x)


def test_guidance_math_properties():
    """pipeline_ltx_video.py:1183-1222 is parity-unpinned (cannot run on CPU); check
    the algebraic properties the formulas imply."""
    g = torch.Generator().manual_seed(0)
    n = torch.randn(3, 24, 16, generator=g)
    # guidance_scale 1 and stg 0 -> text prediction
    out = sched.guidance(n, 3, 1.0, 0.0, 1.0, True, True, False)
    torch.testing.assert_close(out, n[1:2])
    # CFG-star with uncond == text -> alpha 1 -> text
    m = torch.cat([n[1:2], n[1:2], n[1:2]])
    out = sched.guidance(m, 3, 3.0, 1.0, 0.7, True, True, True)
    torch.testing.assert_close(out, n[1:2], rtol=1e-5, atol=1e-6)
    # rescale 1.0 -> std of result equals std of text prediction
    out = sched.guidance(n, 3, 3.0, 1.0, 1.0, True, True, True)
    # (do_rescaling False when all scales are 1.0 in the reference; force the formula)
    out = sched.guidance(n, 3, 3.0, 1.0, 0.999999, True, True, True)
    torch.testing.assert_close(out.std(), n[1].std(), rtol=1e-3, atol=1e-4)


def test_g11_encoder_blocks(golden):
    """Strided CausalConv3d (causal_conv3d.py:33-57) and SpaceToDepthDownsample (:976-1020)."""
    from oracle import vae_encoder as ve
    t, _ = golden("g11_encoder_blocks")
    for name, stride in (("time", (2, 1, 1)), ("space", (1, 2, 2)), ("all", (2, 2, 2))):
        for mode in ("zeros", "replicate"):
            out = ve.strided_causal_conv3d(t["x"], sub(t, f"sconv.{name}.{mode}."), "", stride, mode)
            torch.testing.assert_close(out, t[f"sconv.{name}.{mode}.out"], **TOL)
        blk = dict(stride=stride, group=8 * stride[0] * stride[1] * stride[2] // 16)
        out = ve.space_to_depth_downsample(t["x"], sub(t, f"s2d.{name}.sd."), "", blk, "replicate")
        torch.testing.assert_close(out, t[f"s2d.{name}.out"], **TOL)


def _encoder_case(golden, tag):
    from oracle import vae_encoder as ve
    t, meta = golden(f"g11_encoder_{tag}")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    sd["per_channel_statistics.std-of-means"] = t["per_channel_statistics.std-of-means"]
    sd["per_channel_statistics.mean-of-means"] = t["per_channel_statistics.mean-of-means"]
    tol = dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ve.encoder_forward(sd, cfg, t["x"]), t["encoder"], **tol)
    torch.testing.assert_close(ve.encode(sd, cfg, t["x"]), t["moments"], **tol)
    torch.testing.assert_close(ve.encode(sd, cfg, t["x"][:, :, :1]), t["image_moments"], **tol)
    torch.testing.assert_close(ve.vae_encode(sd, cfg, t["x"]), t["normalized_mean"], **tol)
    out = ve.encode(sd, cfg, t["x_ztile"], use_z_tiling=True, z_sample_size=4)
    torch.testing.assert_close(out, t["moments_ztile"], **tol)
    out = ve.encode(sd, cfg, t["x_hwtile"].float(), use_hw_tiling=True, tile_sample_min_size=128)
    torch.testing.assert_close(out, t["moments_hwtile"], **tol)


def test_g11_encoder_plan_a_strided(golden):
    _encoder_case(golden, "a")


def test_g11_encoder_plan_b_space_to_depth(golden):
    _encoder_case(golden, "b")


def test_g12_conditioning(golden):
    """prepare_conditioning / _get_latent_spatial_position / add_noise_to_image_conditioning_latents /
    per-token timestep / denoising_step with a conditioning mask (pipeline_ltx_video.py:606-629,
    1145-1150, 1309-1342, 1344-1690), run by the reference's own methods."""
    from oracle import conditioning as oc, vae_encoder as ve
    t, meta = golden("g12_conditioning")
    tb, _ = golden("g11_encoder_b")
    cfg = meta["cfg"]
    sd = sub(tb, "sd.")
    torch.testing.assert_close(sd["encoder.conv_out.conv.weight"], t["weights_check"], rtol=0, atol=0)
    sd["per_channel_statistics.std-of-means"] = t["per_channel_statistics.std-of-means"]
    sd["per_channel_statistics.mean-of-means"] = t["per_channel_statistics.mean-of-means"]
    draws = [t[f"noise.{i}"] for i in range(meta["n_prepare_draws"])]
    it = iter(draws)

    def noise_fn(shape):
        n = next(it)
        assert tuple(n.shape) == tuple(shape)
        return n

    items = [oc.ConditioningItem(t["img"], 0, 1.0), oc.ConditioningItem(t["seq"], 8, 0.9),
             oc.ConditioningItem(t["single"], 24, 0.7)]
    lat, pc, mask, n_extra = oc.prepare_conditioning(
        items, t["init_latents"].clone(), meta["F"], meta["H"], meta["W"],
        encode=lambda m: ve.vae_encode(sd, cfg, m), noise_fn=noise_fn)
    assert n_extra == meta["n_extra"] and next(it, None) is None
    tol = dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lat, t["latents"], **tol)
    assert torch.equal(pc, t["pixel_coords"]) and torch.equal(mask, t["mask"])
    lat0, pc0, mask0, n0 = oc.prepare_conditioning(None, t["init_latents"].clone(), meta["F"], meta["H"], meta["W"],
                                                   encode=None, noise_fn=None)
    assert mask0 is None and n0 == 0 and torch.equal(lat0, t["plain.latents"]) and torch.equal(pc0, t["plain.pixel_coords"])
    small = oc.ConditioningItem(torch.zeros(1, 3, 1, 64, 64), 0, 1.0, media_x=32, media_y=None)
    out, lx, ly = oc.get_latent_spatial_position(t["place.in"], small, 128, 160, True)
    assert [lx, ly] == meta["place"] and torch.equal(out, t["place.out"])
    # loop pieces
    ts = t["timesteps"]
    tt = ts[meta["step_index"]]
    noised = oc.add_noise_to_image_conditioning_latents(tt, t["latents"], t["cur_latents"], 0.15, t["mask"],
                                                        t["noise.cond"])
    torch.testing.assert_close(noised, t["noised"], **TOL)
    cur_t = oc.per_token_timestep(tt, t["mask"], 3)
    torch.testing.assert_close(cur_t, t["per_token_timestep"], rtol=0, atol=0)
    stepped = sched.denoising_step(ts, t["noised"], t["v"], cur_t[:1], t["mask"], tt)
    torch.testing.assert_close(stepped, t["stepped"], **TOL)
    # the product's fused form: tokens that move use the GLOBAL dt (see include/ltxmi.h)
    lower = ts[meta["step_index"] + 1]
    fused = torch.where((tt - 1e-6 < 1.0 - t["mask"]).unsqueeze(-1), t["noised"] - (tt - lower) * t["v"], t["noised"])
    torch.testing.assert_close(fused, t["stepped"], **TOL)


def test_g13_latent_upsampler(golden):
    """LatentUpsampler.forward in its four layouts, _upsample_latents and adain_filter_latent
    (latent_upsampler.py:109-149; pipeline_ltx_video.py:1709-1737, 1760-1772)."""
    from oracle import upsampler as up
    t, meta = golden("g13_latent_upsampler")
    tol = dict(rtol=1e-4, atol=2e-5)
    for tag, cfg in meta["cases"].items():
        out = up.latent_upsampler_forward(sub(t, f"{tag}.sd."), cfg, t["latent"])
        assert out.shape == t[f"{tag}.out"].shape, tag
        torch.testing.assert_close(out, t[f"{tag}.out"], **tol)
    ul = up.upsample_latents(sub(t, "d3s.sd."), meta["cases"]["d3s"], t["latent"], t)
    torch.testing.assert_close(ul, t["upsample_latents"], **tol)
    torch.testing.assert_close(up.adain_filter_latent(t["upsample_latents"], t["latent"]), t["adain"], **tol)
    torch.testing.assert_close(up.adain_filter_latent(t["upsample_latents"], t["latent"], 0.5), t["adain_half"], **tol)


def test_g14_pipeline_control(golden):
    """retrieve_timesteps and prepare_latents (pipeline_ltx_video.py:125-198, 632-710)."""
    from oracle import pipeline_ctl as pc
    t, meta = golden("g14_pipeline_control")
    shape = tuple(meta["shape"])
    for i, kw in enumerate(meta["cases"]):
        kw = dict(kw)
        ts = pc.retrieve_timesteps(kw.pop("num_inference_steps", None), shape, **kw)
        torch.testing.assert_close(ts, t[f"ts.{i}"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(pc.prepare_latents(t["latents"], meta["t0"], t["noise.0"], shape), t["prepared"], **TOL)
    torch.testing.assert_close(pc.prepare_latents(None, 1.0, t["noise.1"], shape), t["prepared_none"], **TOL)
    # the per-step guidance tables (:959-1013) are pinned by test_g7_per_step_guidance_tables; here additionally the
    # shipped 13B-dev settings
    gs, stg, rs, skips, do_cfg, do_stg, do_rs = pc.guidance_tables(
        [1.0, 0.99, 0.98, 0.93, 0.85, 0.5, 0.2], [1, 1, 6, 8, 6, 1, 1], [0, 0, 4, 4, 4, 2, 1], [1, 1, 0.5, 0.5, 1, 1, 1],
        [[], [11, 25, 35, 39], [22, 35, 39], [28], [28], [28], [28]],
        guidance_timesteps=[1.0, 0.996, 0.9933, 0.9850, 0.9767, 0.9008, 0.6180])
    # first table entry whose guidance timestep is <= t: mapping = [0, 3, 4, 5, 6, 6, 6]
    assert gs == [0.0, 8, 6, 0.0, 0.0, 0.0, 0.0] and stg == [0, 4, 4, 2, 1, 1, 1] and rs == [1, 0.5, 1, 1, 1, 1, 1]
    assert skips == [[], [28], [28], [28], [28], [28], [28]] and do_cfg and do_stg and do_rs


def _g7_guided(raw, gs, stg, rs, do_cfg, do_stg, do_rs):
    n = 1 + int(do_cfg) + int(do_stg)
    return sched.guidance(raw, n, gs, stg, rs, do_cfg, do_stg, do_rs)


def test_g7_guidance_math_as_run_by_the_reference_call(golden):
    """G7: the guidance block of LTXVideoPipeline.__call__ (:1183-1222), captured from the reference's own __call__
    (config 1: CFG 3 + STG 1 + std-rescale 0.7): what went into denoising_step, from what the transformer returned."""
    t, meta = golden("g7_pipeline_call")
    kw = meta["kwargs"]
    for i in range(2):
        out = _g7_guided(t[f"raw_pred.{i}"], kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"], True, True, True)
        torch.testing.assert_close(out, t[f"guided.{i}"], **TOL)
        # the three conds were fed the same latents; timestep is the scheduler's, one row per cond
        assert torch.equal(t[f"model_in.{i}"][0], t[f"model_in.{i}"][2])
        torch.testing.assert_close(t[f"timestep.{i}"], t["timesteps"][i].expand(3).unsqueeze(-1), rtol=0, atol=0)


def test_g7_per_step_guidance_tables(golden):
    """The per-step tables of __call__ (:959-1013) with list-valued scales, guidance_timesteps and per-step skip lists:
    the oracle's tables must reproduce what the reference fed its own loop (skip masks, and the guided prediction of
    every step from that step's scales -- including the steps whose scales are zeroed)."""
    from oracle import pipeline_ctl as pc
    t, meta = golden("g7_pipeline_call")
    kw = meta["tables_kwargs"]
    ts = t["tables.timesteps"]
    gs, stg, rs, skips, do_cfg, do_stg, do_rs = pc.guidance_tables(
        ts.tolist(), kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"], kw["skip_block_list"],
        guidance_timesteps=kw["guidance_timesteps"])
    assert do_cfg and do_stg and do_rs
    cfg = meta["cfg"]
    for i in range(len(ts)):
        slm = dit.create_skip_layer_mask(cfg["num_layers"], 1, 3, 2, skips[i], torch.float32)
        if f"tables.skip_layer_mask.{i}" in t:
            torch.testing.assert_close(slm, t[f"tables.skip_layer_mask.{i}"], rtol=0, atol=0)
        else:                                   # an empty skip list: the reference passes no mask at all (:171-175)
            assert slm is None and skips[i] == []
        out = _g7_guided(t[f"tables.raw_pred.{i}"], gs[i], stg[i], rs[i], do_cfg, do_stg, do_rs)
        torch.testing.assert_close(out, t[f"tables.guided.{i}"], **TOL)


def test_g11_config1_loop(golden):
    """G11: config 1 end to end (256x256x9, 2 steps, fp32): the oracle's loop -- prepare_latents from the recorded
    noise draw, model, guidance, Euler step, unpatchify -- against the reference's own __call__."""
    from oracle import pipeline_ctl as pc
    t, meta = golden("g7_pipeline_call")
    cfg, kw = meta["cfg"], meta["kwargs"]
    f, h, w = meta["grid"]
    sd = sub(t, "w.")
    shape = (1, cfg["in_channels"], f, h, w)
    ts = pc.retrieve_timesteps(kw["num_inference_steps"], shape)
    torch.testing.assert_close(ts, t["timesteps"], rtol=1e-6, atol=1e-7)
    lat, _ = sched.patchify(pc.prepare_latents(None, 1.0, t["noise"], shape))
    torch.testing.assert_close(lat, t["latents_in.0"], **TOL)
    # the model config of this run has no `causal_temporal_positioning` (default False, as for the 2B 0.9.0 config):
    # prepare_conditioning then builds the pixel coordinates WITHOUT the causal fix (:1530-1538)
    pix = sched.latent_to_pixel_coords(sched.get_latent_coords(f, h, w, 1),
                                       causal_fix=cfg.get("causal_temporal_positioning", False)).to(torch.float32)
    pix[:, 0] = pix[:, 0] * (1.0 / kw["frame_rate"])
    fc = dit.precompute_freqs_cis(pix, cfg, torch.float32)
    emb = torch.cat([t["negative_prompt_embeds"], t["prompt_embeds"], t["prompt_embeds"]])
    msk = torch.cat([t["negative_prompt_attention_mask"], t["prompt_attention_mask"], t["prompt_attention_mask"]])
    skip = dit.create_skip_layer_mask(cfg["num_layers"], 1, 3, 2, kw["skip_block_list"], torch.float32)
    for i, tt in enumerate(ts):
        raw = dit.transformer3d_forward(sd, cfg, torch.cat([lat] * 3), fc, emb, tt.expand(3).unsqueeze(-1),
                                        encoder_attention_mask=msk, latent_shape=(f, h, w), skip_layer_mask=skip,
                                        skip_layer_strategy=dit.ATTENTION_VALUES)
        torch.testing.assert_close(raw, t[f"raw_pred.{i}"], rtol=2e-5, atol=5e-6)
        v = _g7_guided(raw, kw["guidance_scale"], kw["stg_scale"], kw["rescaling_scale"], True, True, True)
        lat = sched.denoising_step(ts, lat, v, tt.expand(1).unsqueeze(-1), None, tt)
        torch.testing.assert_close(lat, t[f"latents_out.{i}"], rtol=2e-5, atol=5e-6)
    torch.testing.assert_close(sched.unpatchify(lat, f, h, w), t["out_latents"], rtol=2e-5, atol=5e-6)


def test_g10c_decoder_block_variants(golden):
    """compress_time / compress_space upsamplers and inject_noise resnet blocks through the reference's own
    Decoder.forward (causal_video_autoencoder.py:671-684, 1183-1195, 1230, 1246), with the recorded noise draws."""
    t, meta = golden("g10c_decoder_variants")
    cfg = meta["cfg"]
    sd = sub(t, "sd.")
    noises = [t[f"noise.{i}"] for i in range(len(meta["noise_shapes"]))]
    out = vae.decoder_forward(sd, cfg, t["z"], t["timestep"], noises=noises)
    torch.testing.assert_close(out, t["out"], **TOL)
    # the noise really acts
    quiet = vae.decoder_forward(sd, cfg, t["z"], t["timestep"], noises=[torch.zeros_like(n) for n in noises])
    assert (quiet - t["out"]).abs().max() > 1e-3


def test_g6_scheduler_stochastic_step(golden):
    """RectifiedFlowScheduler.step(stochastic_sampling=True) (rf.py:368-373) with the recorded noise draw."""
    t, _ = golden("g6_scheduler")
    tsched = t["step.timesteps"]
    for name, tt in (("global", tsched[2][None, None].expand(1, 24)), ("per_token", t["step.tok_t"])):
        out = sched.scheduler_step(tsched, t["step.v"], tt, t["step.sample"], stochastic_noise=t[f"step.stochastic_{name}_noise"])
        torch.testing.assert_close(out, t[f"step.stochastic_{name}"], **TOL)


def test_g0_timestep_embedding(golden):
    """G0: oracle/leaves.py::get_timestep_embedding against the reference's own in-repo copy
    (ltx_video/models/transformers/embeddings.py:10-50) -- the one diffusers-shaped leaf that IS pinned."""
    from oracle import leaves
    t, _ = golden("g0_timestep_embedding")
    ts = t["timesteps"]
    assert torch.equal(leaves.get_timestep_embedding(ts, 256, flip_sin_to_cos=True, downscale_freq_shift=0.0),
                       t["emb_256_flip_shift0"])
    assert torch.equal(leaves.get_timestep_embedding(ts, 64), t["emb_64_defaults"])
    assert torch.equal(leaves.get_timestep_embedding(ts, 33, False, 1, 2.0, 1000), t["emb_33_odd_scale2"])


def g15_case(golden):
    """Everything the G15 replay needs, rebuilt from the manifest: the DiT weights are G7's, the VAE decoder and the latent
    upsampler come from the oracle's seeded initialisers (fingerprints checked, so an RNG drift fails loudly here and not
    as a parity miss), the prompts go through the same fake T5 as in the generator."""
    from oracle import upsampler as ou, vae as ov
    from fake_t5 import FakeTextEncoder, FakeTokenizer
    t, meta = golden("g15_multiscale_call")
    w, _ = golden("g7_pipeline_call")
    sd = sub(w, "w.")
    vsd = ov.init_state_dict(meta["vae_cfg"], seed=meta["vae_seed"])
    usd = ou.init_state_dict(meta["upsampler_cfg"], seed=meta["upsampler_seed"])

    def fingerprint(d):
        return float(sum(v.double().abs().sum() for v in d.values()))

    assert abs(fingerprint(vsd) - meta["vae_fingerprint"]) < 1e-6 * meta["vae_fingerprint"], "seeded VAE weights drifted"
    assert abs(fingerprint(usd) - meta["upsampler_fingerprint"]) < 1e-6 * meta["upsampler_fingerprint"]
    tok, enc = FakeTokenizer(), FakeTextEncoder(meta["dit_cfg"]["caption_channels"], seed=meta["text_encoder_seed"]).eval()
    return t, meta, sd, vsd, usd, tok, enc


def test_g15_multiscale_call_with_ltxv_kwargs(golden):
    """G15: the reference's own LTXMultiScalePipeline.__call__ run with the keyword arguments of ltxv.py:420-445 (YAML spread,
    string prompts, output_type "pt", callback).  (1) the fake T5 + the reference's encode_prompt arithmetic reproduce the
    recorded embeddings; (2) the oracle's two-pass restatement lands on the recorded pass-1 latents, upsampled latents and
    final video."""
    from oracle import pipeline_ctl as pc
    t, meta, sd, vsd, usd, tok, enc = g15_case(golden)
    call, cfgp = meta["call"], meta["pipeline_config"]
    with torch.no_grad():
        ti = tok([call["prompt"].strip()], padding="max_length", max_length=256, truncation=True)
        pos = enc(ti.input_ids, attention_mask=ti.attention_mask)[0]
        ni = tok([call["negative_prompt"].strip()], padding="max_length", max_length=256, truncation=True)
        neg = enc(ni.input_ids, attention_mask=ni.attention_mask)[0]
    assert torch.equal(pos, t["prompt_embeds"]) and torch.equal(neg, t["negative_prompt_embeds"])
    assert torch.equal(ti.attention_mask.float(), t["prompt_attention_mask"])
    assert torch.equal(ni.attention_mask.float(), t["negative_prompt_attention_mask"])
    video, lat1, up = pc.multiscale_call(
        sd, meta["dit_cfg"], vsd, meta["vae_cfg"], usd, meta["upsampler_cfg"], pos, neg, t["prompt_attention_mask"],
        t["negative_prompt_attention_mask"], call["height"], call["width"], call["num_frames"], call["frame_rate"],
        cfgp["downscale_factor"], cfgp["first_pass"], cfgp["second_pass"], call["num_inference_steps1"],
        call["num_inference_steps2"], t["noise.0"], t["noise.1"], t["decode_noise"], cfgp["decode_timestep"],
        cfgp["decode_noise_scale"], stats=vsd)
    torch.testing.assert_close(lat1, t["pass1_latents"], rtol=5e-5, atol=1e-5)
    torch.testing.assert_close(up, t["upsampled"], rtol=5e-5, atol=1e-5)
    assert video.shape == t["images"].shape == tuple(meta["images_shape"])
    assert float(t["images"].std()) > 0.02                       # a real picture, not the clamp's floor / ceiling
    torch.testing.assert_close(video, t["images"], rtol=1e-4, atol=1e-4)
