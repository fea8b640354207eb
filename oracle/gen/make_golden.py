"""Generate the golden vectors in tests/golden/ by running the REFERENCE's own code.

BUILD-CONTAINER-ONLY: imports the reference's modules from /root/reference through
oracle/gen/ref_shims.py, runs seeded tiny cases on CPU, and stores inputs, weights
and outputs as plain tensors (.safetensors) plus a JSON manifest.  Only data is
written -- no reference source, bytecode or pickled objects.

    python -m oracle.gen.make_golden          # from the repo root

Cases: G1 precompute_freqs_cis, G2 apply_rotary_emb, G3 attention
processor (self / cross+mask / STG values), G4 BasicTransformerBlock (per-batch and
per-frame timestep), G5 Transformer3DModel.forward (fp32 + bf16 twin), G6 scheduler,
G8 CausalConv3d, G9 ResnetBlock3D / DepthToSpaceUpsample / PixelNorm / (un)patchify,
G10 Decoder.forward + decode (plain, z-tiled, hw-tiled) + vae_decode for both
decoder block plans.
"""
import json
import os
import sys
import types

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gen import ref_shims  # noqa: E402

ref_shims.install()

import ltx_video.models.transformers.transformer3d as ref_t3  # noqa: E402
import ltx_video.models.transformers.attention as ref_attn  # noqa: E402
import ltx_video.models.autoencoders.causal_video_autoencoder as ref_cva  # noqa: E402
import ltx_video.models.autoencoders.causal_conv3d as ref_cc3  # noqa: E402
import ltx_video.models.autoencoders.pixel_norm as ref_pn  # noqa: E402
import ltx_video.models.autoencoders.vae_encode as ref_ve  # noqa: E402
import ltx_video.schedulers.rf as ref_rf  # noqa: E402
import ltx_video.models.transformers.symmetric_patchifier as ref_sp  # noqa: E402
from ltx_video.utils.skip_layer_strategy import SkipLayerStrategy  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MANIFEST = {}


def save(name, tensors, meta=None):
    t = {k: v.detach().clone().contiguous() for k, v in tensors.items()}
    save_file(t, os.path.join(OUT, name + ".safetensors"))
    MANIFEST[name] = meta or {}
    print(f"  {name}: {len(t)} tensors, {sum(v.numel() * v.element_size() for v in t.values()) / 1e6:.2f} MB")


TINY_DIT = dict(
    num_attention_heads=2, attention_head_dim=32, in_channels=16, out_channels=16, num_layers=2,
    cross_attention_dim=64, caption_channels=32, attention_bias=True, activation_fn="gelu-approximate",
    norm_elementwise_affine=False, norm_eps=1e-6, qk_norm="rms_norm", standardization_norm="rms_norm",
    adaptive_norm="single_scale_shift", positional_embedding_type="rope",
    positional_embedding_theta=10000.0, positional_embedding_max_pos=[20, 2048, 2048],
    timestep_scale_multiplier=1000,
)


def build_dit(cfg, seed):
    torch.manual_seed(seed)
    model = ref_t3.Transformer3DModel(**cfg).eval()
    # move q/k norm weights off their all-ones init so the weight path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("q_norm.weight") or n.endswith("k_norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
    return model


def coords(f, h, w, b, fps=25.0):
    pat = ref_sp.SymmetricPatchifier(patch_size=1)
    lc = pat.get_latent_coords(f, h, w, b, "cpu")
    pc = ref_ve.latent_to_pixel_coords_from_factors(lc, (8, 32, 32), causal_fix=True).to(torch.float32)
    pc[:, 0] = pc[:, 0] * (1.0 / fps)
    return lc, pc


@torch.no_grad()
def g1_g2():
    print("G1/G2 rope")
    t = {}
    meta = {"cases": []}
    for tag, (heads, dh, grid) in {"d48": (2, 24, (2, 4, 4)), "d64": (2, 32, (3, 4, 6)),
                                   "d2048": (32, 64, (2, 2, 3))}.items():
        cfg = dict(TINY_DIT, num_attention_heads=heads, attention_head_dim=dh, num_layers=1,
                   cross_attention_dim=heads * dh)
        model = build_dit(cfg, 0)
        lc, pc = coords(*grid, 2)
        cos, sin = model.precompute_freqs_cis(pc)
        t[f"{tag}.latent_coords"] = lc
        t[f"{tag}.indices_grid"] = pc
        t[f"{tag}.cos"], t[f"{tag}.sin"] = cos, sin
        x = torch.randn(2, pc.shape[-1], heads * dh, generator=torch.Generator().manual_seed(3))
        t[f"{tag}.x"] = x
        t[f"{tag}.rot"] = ref_attn.Attention.apply_rotary_emb(x, (cos, sin))
        meta["cases"].append(dict(tag=tag, heads=heads, dh=dh, grid=grid))
    save("g1_rope", t, meta)


def dit_inputs(cfg, f, h, w, B, T, seed, per_token_timestep=False):
    g = torch.Generator().manual_seed(seed)
    N = f * h * w
    x = torch.randn(B, N, cfg["in_channels"], generator=g)
    enc = torch.randn(B, T, cfg["caption_channels"], generator=g)
    mask = torch.ones(B, T)
    mask[:, T - 3:] = 0
    mask[0, T - 5:] = 0
    if per_token_timestep:
        ts = torch.full((B, N), 0.7)
        ts[:, : h * w] = 0.0            # first latent frame is hard-conditioned (i2v)
    else:
        ts = torch.full((B, 1), 0.7)
    return x, enc, mask, ts


@torch.no_grad()
def g3_g4_g5():
    print("G3/G4/G5 DiT")
    cfg = TINY_DIT
    f, h, w, B, T = 2, 3, 4, 3, 8
    model = build_dit(cfg, 10)
    sd = {k: v for k, v in model.state_dict().items()}
    _, pc = coords(f, h, w, 1)
    freqs = model.precompute_freqs_cis(pc)
    holder = types.SimpleNamespace(_interrupt=False)
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]

    # ---- G3: attention processors on the first block
    blk = model.transformer_blocks[0]
    g = torch.Generator().manual_seed(20)
    hs = torch.randn(B, f * h * w, D, generator=g)
    ctx = torch.randn(B, T, D, generator=g)
    bias = torch.zeros(B, 1, T)
    bias[:, :, T - 3:] = -10000.0
    t3 = {"hs": hs, "ctx": ctx, "bias": bias, "cos": freqs[0], "sin": freqs[1]}
    t3["self"] = blk.attn1([hs.clone()], freqs_cis=freqs)
    t3["cross"] = blk.attn2([hs.clone()], freqs_cis=freqs, encoder_hidden_states=ctx, attention_mask=bias)
    slm = torch.tensor([1.0, 1.0, 0.0])
    t3["slm"] = slm
    t3["self_stg_values"] = blk.attn1([hs.clone()], freqs_cis=freqs, skip_layer_mask=slm,
                                      skip_layer_strategy=SkipLayerStrategy.AttentionValues)
    t3["self_stg_skip"] = blk.attn1([hs.clone()], freqs_cis=freqs, skip_layer_mask=slm,
                                    skip_layer_strategy=SkipLayerStrategy.AttentionSkip)
    one = torch.tensor([0.0])
    t3["self_stg_values_b1"] = blk.attn1([hs[:1].clone()], freqs_cis=freqs, skip_layer_mask=one,
                                         skip_layer_strategy=SkipLayerStrategy.AttentionValues)
    for k, v in sd.items():
        if k.startswith("transformer_blocks.0."):
            t3["sd." + k] = v
    save("g3_attention", t3, dict(cfg=cfg, grid=(f, h, w), B=B, T=T))

    # ---- G4: one block, per-batch and per-frame timestep tables
    t4 = {"hs": hs, "ctx": ctx, "bias": bias, "cos": freqs[0], "sin": freqs[1]}
    temb_b = torch.randn(B, 1, 6 * D, generator=g) * 0.5
    temb_f = torch.randn(B, f, 6 * D, generator=g) * 0.5
    t4["temb_b"], t4["temb_f"] = temb_b, temb_f
    t4["out_b"] = blk(hs.clone(), freqs_cis=freqs, encoder_hidden_states=ctx,
                      encoder_attention_mask=bias, timestep=temb_b)
    t4["out_f"] = blk(hs.clone(), freqs_cis=freqs, encoder_hidden_states=ctx,
                      encoder_attention_mask=bias, timestep=temb_f)
    t4["out_b_stg"] = blk(hs.clone(), freqs_cis=freqs, encoder_hidden_states=ctx,
                          encoder_attention_mask=bias, timestep=temb_b, skip_layer_mask=slm,
                          skip_layer_strategy=SkipLayerStrategy.AttentionValues)
    t4["out_b_tb"] = blk(hs.clone(), freqs_cis=freqs, encoder_hidden_states=ctx,
                         encoder_attention_mask=bias, timestep=temb_b, skip_layer_mask=slm,
                         skip_layer_strategy=SkipLayerStrategy.TransformerBlock)
    for k, v in sd.items():
        if k.startswith("transformer_blocks.0."):
            t4["sd." + k] = v
    save("g4_block", t4, dict(cfg=cfg, grid=(f, h, w), B=B, T=T))

    # ---- G5: whole model
    t5 = {"indices_grid": pc}
    x, enc, mask, ts = dit_inputs(cfg, f, h, w, B, T, 30)
    skip = model.create_skip_layer_mask(1, 3, 2, [1])
    t5.update({"x": x, "enc": enc, "mask": mask, "ts": ts, "skip_layer_mask": skip})
    kw = dict(freqs_cis=freqs, encoder_hidden_states=enc, encoder_attention_mask=mask,
              latent_shape=(f, h, w), ltxv_model=holder, return_dict=False)
    t5["out"] = model(x.clone(), timestep=ts, **kw)[0]
    t5["out_stg"] = model(x.clone(), timestep=ts, skip_layer_mask=skip,
                          skip_layer_strategy=SkipLayerStrategy.AttentionValues, **kw)[0]
    _, _, _, ts_tok = dit_inputs(cfg, f, h, w, B, T, 30, per_token_timestep=True)
    t5["ts_tok"] = ts_tok
    t5["out_tok"] = model(x.clone(), timestep=ts_tok, **kw)[0]
    for k, v in sd.items():
        t5["sd." + k] = v
    # bf16 twin: the reference's eager path in bf16 (weights, activations, tables)
    mb = build_dit(cfg, 10).to(torch.bfloat16)
    fb = mb.precompute_freqs_cis(pc)
    t5["bf16.cos"], t5["bf16.sin"] = fb
    t5["bf16.out"] = mb(x.to(torch.bfloat16), freqs_cis=fb, encoder_hidden_states=enc.to(torch.bfloat16),
                        encoder_attention_mask=mask, timestep=ts, latent_shape=(f, h, w),
                        ltxv_model=holder, return_dict=False)[0]
    t5["bf16.out_stg"] = mb(x.to(torch.bfloat16), freqs_cis=fb, encoder_hidden_states=enc.to(torch.bfloat16),
                            encoder_attention_mask=mask, timestep=ts, latent_shape=(f, h, w),
                            skip_layer_mask=skip.to(torch.bfloat16),
                            skip_layer_strategy=SkipLayerStrategy.AttentionValues,
                            ltxv_model=holder, return_dict=False)[0]
    save("g5_transformer", t5, dict(cfg=cfg, grid=(f, h, w), B=B, T=T, skip_blocks=[1]))


@torch.no_grad()
def g6():
    print("G6 scheduler")
    t = {}
    shapes = {"c1": (1, 128, 2, 8, 8), "c2": (1, 128, 13, 16, 24), "c3": (1, 128, 16, 22, 38),
              "c5": (1, 128, 33, 23, 40), "small": (1, 128, 2, 3, 4)}
    for tag, shp in shapes.items():
        for steps in (2, 8, 40):
            s = ref_rf.RectifiedFlowScheduler(num_train_timesteps=1000, shifting="SD3",
                                              base_resolution=None, target_shift_terminal=0.1)
            s.set_timesteps(steps, samples_shape=torch.Size(shp), device="cpu")
            t[f"{tag}.steps{steps}"] = s.timesteps
    s = ref_rf.RectifiedFlowScheduler(num_train_timesteps=1000, shifting="SD3", base_resolution=None,
                                      target_shift_terminal=0.1)
    s.set_timesteps(8, samples_shape=torch.Size(shapes["small"]), device="cpu")
    g = torch.Generator().manual_seed(5)
    sample = torch.randn(1, 24, 16, generator=g)
    v = torch.randn(1, 24, 16, generator=g)
    t["step.timesteps"] = s.timesteps
    t["step.sample"], t["step.v"] = sample, v
    t["step.global"] = s.step(v, s.timesteps[2], sample, return_dict=False)[0]
    tok = s.timesteps[2].expand(1, 24).clone()
    tok[:, :12] = 0.0
    tok[:, 12:16] = 0.31
    t["step.tok_t"] = tok
    t["step.per_token"] = s.step(v, tok, sample, return_dict=False)[0]
    # stochastic sampling (rf.py:368-373): x0 = x - t v, re-noised to the next timestep with torch.randn_like(sample)
    # (unseeded in the reference: made reproducible with manual_seed and recorded by drawing it again)
    for name, tt in (("global", s.timesteps[2][None, None].expand(1, 24).clone()), ("per_token", tok)):
        torch.manual_seed(7)
        t[f"step.stochastic_{name}"] = s.step(v, tt, sample, return_dict=False, stochastic_sampling=True)[0]
        torch.manual_seed(7)
        t[f"step.stochastic_{name}_noise"] = torch.randn_like(sample)
    save("g6_scheduler", t, dict(shapes={k: list(v) for k, v in shapes.items()}))


TINY_VAE_A = {  # OURS_VAE_CONFIG block plan (diffusers_config_mapping.py:106-130), shrunk
    "_class_name": "CausalVideoAutoencoder", "dims": 3, "in_channels": 3, "out_channels": 3,
    "latent_channels": 8,
    "blocks": [["res_x", 1], ["compress_all", 1], ["res_x_y", 1], ["res_x", 1], ["compress_all", 1],
               ["res_x_y", 1], ["res_x", 1], ["compress_all", 1], ["res_x", 1], ["res_x", 1]],
    "scaling_factor": 1.0, "norm_layer": "pixel_norm", "patch_size": 4, "latent_log_var": "uniform",
    "use_quant_conv": False, "causal_decoder": False,
    "encoder_base_channels": 8, "decoder_base_channels": 8,
}


def tiny_vae_b():
    cfg = ref_cva.create_video_autoencoder_demo_config(latent_channels=8)
    cfg["encoder_base_channels"] = 8
    cfg["decoder_base_channels"] = 8
    return cfg


def jsonable(cfg):
    return json.loads(json.dumps(cfg))


@torch.no_grad()
def g8_g9():
    print("G8/G9 conv + blocks")
    g = torch.Generator().manual_seed(40)
    t = {}
    x = torch.randn(2, 6, 4, 5, 7, generator=g)
    t["x"] = x
    for mode in ("zeros", "replicate"):
        torch.manual_seed(41)
        conv = ref_cc3.CausalConv3d(6, 10, kernel_size=3, spatial_padding_mode=mode)
        t[f"conv.{mode}.conv.weight"], t[f"conv.{mode}.conv.bias"] = conv.conv.weight, conv.conv.bias
        t[f"conv.{mode}.causal"] = conv(x, causal=True)
        t[f"conv.{mode}.noncausal"] = conv(x, causal=False)
    t["pixel_norm"] = ref_pn.PixelNorm()(x)
    xp = torch.randn(2, 3, 8, 64, 64, generator=g)[:, :, :2, :16, :16].contiguous()
    t["patch.x"] = xp
    t["patch.patchified"] = ref_cva.patchify(xp, patch_size_hw=4, patch_size_t=1)
    t["patch.roundtrip"] = ref_cva.unpatchify(t["patch.patchified"], patch_size_hw=4, patch_size_t=1)

    torch.manual_seed(42)
    res = ref_cva.ResnetBlock3D(dims=3, in_channels=8, out_channels=8, eps=1e-6, norm_layer="pixel_norm",
                                timestep_conditioning=True, spatial_padding_mode="replicate").eval()
    xr = torch.randn(2, 8, 3, 4, 5, generator=g)
    temb = torch.randn(2, 32, 1, 1, 1, generator=g) * 0.5
    t["res.x"], t["res.temb"] = xr, temb
    t["res.out"] = res(xr, causal=False, timestep=temb)
    for k, v in res.state_dict().items():
        t["res.sd." + k] = v
    torch.manual_seed(43)
    resxy = ref_cva.ResnetBlock3D(dims=3, in_channels=8, out_channels=4, eps=1e-6, norm_layer="pixel_norm",
                                  timestep_conditioning=False, spatial_padding_mode="zeros").eval()
    t["resxy.out"] = resxy(xr, causal=False)
    for k, v in resxy.state_dict().items():
        t["resxy.sd." + k] = v
    torch.manual_seed(44)
    up = ref_cva.DepthToSpaceUpsample(dims=3, in_channels=8, stride=(2, 2, 2), residual=True,
                                      out_channels_reduction_factor=2, spatial_padding_mode="replicate").eval()
    t["up.out"] = up(xr, causal=False)
    for k, v in up.state_dict().items():
        t["up.sd." + k] = v
    torch.manual_seed(45)
    up2 = ref_cva.DepthToSpaceUpsample(dims=3, in_channels=8, stride=(2, 2, 2), residual=False,
                                       out_channels_reduction_factor=1, spatial_padding_mode="zeros").eval()
    t["up2.out"] = up2(xr, causal=False)
    for k, v in up2.state_dict().items():
        t["up2.sd." + k] = v
    save("g8_conv_blocks", t)


@torch.no_grad()
def g10():
    print("G10 decoder")
    for tag, cfg in (("a", jsonable(TINY_VAE_A)), ("b", jsonable(tiny_vae_b()))):
        torch.manual_seed(50)
        vae = ref_cva.CausalVideoAutoencoder.from_config(json.loads(json.dumps(cfg))).eval()
        tcond = cfg.get("timestep_conditioning", False)
        g = torch.Generator().manual_seed(51)
        C = cfg["latent_channels"]
        std = 0.5 + torch.rand(C, generator=g)
        mean = 0.2 * torch.randn(C, generator=g)
        vae.register_buffer("std_of_means", std)
        vae.register_buffer("mean_of_means", mean)
        t = {"per_channel_statistics.std-of-means": std, "per_channel_statistics.mean-of-means": mean}
        for k, v in vae.state_dict().items():
            if k.startswith("decoder."):
                t["sd." + k] = v
        z = torch.randn(1, C, 2, 2, 2, generator=g)
        ts = torch.tensor([0.05]) if tcond else None
        t["z"] = z
        F_, H, W = z.shape[2:]
        tgt = (1, 3, (F_ - 1) * 8 + 1, H * 32, W * 32)
        t["decode"] = vae.decode(z, return_dict=False, target_shape=tgt, timestep=ts)[0]
        t["vae_decode"] = ref_ve.vae_decode(z, vae, is_video=True, vae_per_channel_normalize=True, timestep=ts)
        # z-tiling (tile = 4+1 latent frames, vae.py:365-402)
        zz = torch.randn(1, C, 7, 1, 1, generator=g)
        t["z_ztile"] = zz
        vae.enable_z_tiling(4)
        t["decode_ztile"] = vae.decode(zz, return_dict=False, target_shape=tgt, timestep=ts)[0]
        vae.disable_z_tiling()
        t["decode_ztile_ref_untiled"] = vae.decode(zz, return_dict=False, target_shape=tgt, timestep=ts)[0]
        # hw-tiling with a 64-px tile (tile_latent_min_size = 2), vae.py:223-263
        zh = torch.randn(1, C, 1, 3, 4, generator=g)
        t["z_hwtile"] = zh
        vae.set_tiling_params(sample_size=64, overlap_factor=0.25)
        vae.enable_hw_tiling()
        t["decode_hwtile"] = vae.decode(zh, return_dict=False, target_shape=tgt, timestep=ts)[0]
        vae.disable_hw_tiling()
        if tcond:
            t["timestep"] = ts
        save(f"g10_decoder_{tag}", t, dict(cfg=cfg))


@torch.no_grad()
def g11():
    """Encoder side: strided CausalConv3d, SpaceToDepthDownsample, Encoder.forward, _encode,
    z-/hw-tiled encode and normalize_latents for both block plans.  The diffusers leaves
    (DiagonalGaussianDistribution / AutoencoderKLOutput) are absent: encode() is run with them
    replaced by identities so that it returns the moments tensor it would wrap."""
    print("G11 encoder")
    import ltx_video.models.autoencoders.vae as ref_vae
    ref_vae.DiagonalGaussianDistribution = lambda moments: moments
    ref_vae.AutoencoderKLOutput = lambda latent_dist: latent_dist
    g = torch.Generator().manual_seed(60)
    t = {}
    xs = torch.randn(2, 8, 5, 6, 8, generator=g)
    t["x"] = xs
    for name, stride in (("time", (2, 1, 1)), ("space", (1, 2, 2)), ("all", (2, 2, 2))):
        for mode in ("zeros", "replicate"):
            torch.manual_seed(61)
            conv = ref_cc3.CausalConv3d(8, 12, kernel_size=3, stride=stride, spatial_padding_mode=mode)
            t[f"sconv.{name}.{mode}.conv.weight"], t[f"sconv.{name}.{mode}.conv.bias"] = conv.conv.weight, conv.conv.bias
            t[f"sconv.{name}.{mode}.out"] = conv(xs, causal=True)
        torch.manual_seed(62)
        s2d = ref_cva.SpaceToDepthDownsample(dims=3, in_channels=8, out_channels=16, stride=stride,
                                             spatial_padding_mode="replicate").eval()
        xin = xs if stride[0] == 1 else xs[:, :, :5]          # T=5 -> 6 frames after the duplicated first one
        t[f"s2d.{name}.out"] = s2d(xin)
        for k, v in s2d.state_dict().items():
            t[f"s2d.{name}.sd.{k}"] = v
    save("g11_encoder_blocks", t)

    for tag, cfg in (("a", jsonable(TINY_VAE_A)), ("b", jsonable(tiny_vae_b()))):
        torch.manual_seed(63)
        vae = ref_cva.CausalVideoAutoencoder.from_config(json.loads(json.dumps(cfg))).eval()
        C = cfg["latent_channels"]
        std = 0.5 + torch.rand(C, generator=g)
        mean = 0.2 * torch.randn(C, generator=g)
        vae.register_buffer("std_of_means", std)
        vae.register_buffer("mean_of_means", mean)
        t = {"per_channel_statistics.std-of-means": std, "per_channel_statistics.mean-of-means": mean}
        for k, v in vae.state_dict().items():
            if k.startswith("encoder."):
                t["sd." + k] = v
        x = torch.rand(1, 3, 9, 64, 32, generator=g) * 2 - 1
        t["x"] = x
        t["encoder"] = vae.encoder(x)
        t["moments"] = vae.encode(x, return_dict=False)[0]
        t["image_moments"] = vae.encode(x[:, :, :1], return_dict=False)[0]
        t["normalized_mean"] = ref_ve.normalize_latents(t["moments"][:, :C], vae, vae_per_channel_normalize=True)
        xz = torch.rand(1, 3, 41, 32, 32, generator=g) * 2 - 1
        t["x_ztile"] = xz
        vae.enable_z_tiling(4)
        t["moments_ztile"] = vae.encode(xz, return_dict=False)[0]
        vae.disable_z_tiling()
        xh = torch.rand(1, 3, 9, 128, 160, generator=g).to(torch.bfloat16)    # stored as bf16 (size)
        t["x_hwtile"] = xh
        vae.set_tiling_params(sample_size=128, overlap_factor=0.25)
        vae.enable_hw_tiling()
        t["moments_hwtile"] = vae.encode(xh.float(), return_dict=False)[0]
        vae.disable_hw_tiling()
        save(f"g11_encoder_{tag}", t, dict(cfg=cfg))


@torch.no_grad()
def g12():
    """Conditioning-token assembly and the masked denoising step, run through the reference's own
    LTXVideoPipeline methods on an instance created WITHOUT __init__ (no text encoder / transformer
    needed for these methods).  The diffusers leaves are stand-ins: DiagonalGaussianDistribution.sample()
    returns the mean (the reference draws unseeded noise there), randn_tensor is torch.randn and its
    draws are recorded so that the oracle/product can be fed the same noise."""
    print("G12 conditioning")
    ref_shims.install_pipeline_leaves()
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl
    import ltx_video.models.autoencoders.vae as ref_vae

    class _Dist:
        def __init__(self, moments):
            self.mean = moments[:, : moments.shape[1] // 2]

        def sample(self):
            return self.mean

    ref_vae.DiagonalGaussianDistribution = _Dist
    ref_vae.AutoencoderKLOutput = lambda latent_dist: types.SimpleNamespace(latent_dist=latent_dist)
    draws = []

    def logged_randn(shape, generator=None, device=None, dtype=None, layout=None):
        n = ref_shims.randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        draws.append(n)
        return n

    ref_pl.randn_tensor = logged_randn

    cfg = jsonable(tiny_vae_b())
    torch.manual_seed(63)          # same construction + seed as g11 "b": the encoder weights live in g11_encoder_b
    vae = ref_cva.CausalVideoAutoencoder.from_config(json.loads(json.dumps(cfg))).eval()
    g = torch.Generator().manual_seed(71)
    C = cfg["latent_channels"]
    std = 0.5 + torch.rand(C, generator=g)
    mean = 0.2 * torch.randn(C, generator=g)
    vae.register_buffer("std_of_means", std)
    vae.register_buffer("mean_of_means", mean)
    t = {"per_channel_statistics.std-of-means": std, "per_channel_statistics.mean-of-means": mean}
    t["weights_check"] = vae.state_dict()["encoder.conv_out.conv.weight"]
    pipe = object.__new__(ref_pl.LTXVideoPipeline)
    pipe.vae = vae
    pipe.patchifier = ref_sp.SymmetricPatchifier(patch_size=1)
    pipe.vae_scale_factor = 32
    pipe.transformer = types.SimpleNamespace(config=types.SimpleNamespace(causal_temporal_positioning=True),
                                             use_tpu_flash_attention=False)
    H, W, F_ = 64, 96, 33
    img = torch.rand(1, 3, 1, H, W, generator=g) * 2 - 1
    seq = torch.rand(1, 3, 17, H, W, generator=g) * 2 - 1
    single = torch.rand(1, 3, 1, H, W, generator=g) * 2 - 1
    t["img"], t["seq"], t["single"] = img, seq, single
    items = [ref_pl.ConditioningItem(img, 0, 1.0), ref_pl.ConditioningItem(seq, 8, 0.9),
             ref_pl.ConditioningItem(single, 24, 0.7)]
    init = torch.randn(1, C, 5, 2, 3, generator=g)
    t["init_latents"] = init.clone()
    gen = torch.Generator().manual_seed(72)
    lat, pc, mask, n_extra = pipe.prepare_conditioning(items, init, F_, H, W, vae_per_channel_normalize=True,
                                                       generator=gen)
    t["latents"], t["pixel_coords"], t["mask"] = lat, pc, mask
    for i, n in enumerate(draws):
        t[f"noise.{i}"] = n
    n_draws = len(draws)
    # no conditioning items: plain patchify + coords
    lat0, pc0, mask0, n0 = pipe.prepare_conditioning(None, t["init_latents"].clone(), F_, H, W, vae_per_channel_normalize=True)
    assert mask0 is None and n0 == 0
    t["plain.latents"], t["plain.pixel_coords"] = lat0, pc0
    # spatial placement of a smaller first-frame item (strip_latent_border)
    small = ref_pl.ConditioningItem(torch.zeros(1, 3, 1, 64, 64), 0, 1.0, media_x=32, media_y=None)
    zl = torch.randn(1, C, 1, 2, 2, generator=g)
    out, lx, ly = pipe._get_latent_spatial_position(zl, small, 128, 160, strip_latent_border=True)
    t["place.in"], t["place.out"] = zl, out.contiguous()
    # image-conditioning noise + masked denoising step with the reference scheduler
    sch = ref_rf.RectifiedFlowScheduler(num_train_timesteps=1000, shifting="SD3", base_resolution=None,
                                        target_shift_terminal=0.1)
    sch.set_timesteps(8, samples_shape=torch.Size((1, C, 5, 2, 3)), device="cpu")
    pipe.scheduler = sch
    t["timesteps"] = sch.timesteps
    tt = sch.timesteps[3]
    gen2 = torch.Generator().manual_seed(73)
    cur = torch.randn(lat.shape, generator=g)
    t["cur_latents"] = cur
    noised = ref_pl.LTXVideoPipeline.add_noise_to_image_conditioning_latents(tt, lat, cur, 0.15, mask, gen2)
    t["noised"], t["noise.cond"] = noised, draws[n_draws]
    v = torch.randn(lat.shape, generator=g)
    t["v"] = v
    num_conds = 3
    cur_t = tt[None].expand(num_conds).unsqueeze(-1)
    cur_t = torch.min(cur_t, 1.0 - torch.cat([mask] * num_conds))
    t["per_token_timestep"] = cur_t
    t["stepped"] = pipe.denoising_step(noised, v, cur_t[:1], mask, tt, {})
    save("g12_conditioning", t, dict(cfg=cfg, n_extra=int(n_extra), place=[int(lx), int(ly)], H=H, W=W, F=F_,
                                     step_index=3, n_prepare_draws=n_draws))


@torch.no_grad()
def g13():
    """LatentUpsampler.forward (dims 3 spatial = the shipped configuration, dims 2, dims 3 spatial+temporal,
    dims 3 temporal), adain_filter_latent and LTXMultiScalePipeline._upsample_latents."""
    print("G13 latent upsampler")
    ref_shims.install_pipeline_leaves()
    import ltx_video.models.autoencoders.latent_upsampler as ref_lu
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl
    g = torch.Generator().manual_seed(80)
    t, meta = {}, {"cases": {}}
    lat = torch.randn(2, 8, 3, 4, 5, generator=g)
    t["latent"] = lat
    cases = {"d3s": dict(dims=3, spatial_upsample=True, temporal_upsample=False),
             "d2s": dict(dims=2, spatial_upsample=True, temporal_upsample=False),
             "d3st": dict(dims=3, spatial_upsample=True, temporal_upsample=True),
             "d3t": dict(dims=3, spatial_upsample=False, temporal_upsample=True)}
    for i, (tag, kw) in enumerate(cases.items()):
        cfg = dict(in_channels=8, mid_channels=32, num_blocks_per_stage=1, **kw)
        torch.manual_seed(81 + i)
        m = ref_lu.LatentUpsampler.from_config(cfg).eval()
        for n, p in m.named_parameters():          # move the GroupNorm affine off its identity init
            if "norm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for k, v in m.state_dict().items():
            t[f"{tag}.sd.{k}"] = v
        t[f"{tag}.out"] = m(lat)
        meta["cases"][tag] = cfg
        if tag == "d3s":
            C = 8
            std = 0.5 + torch.rand(C, generator=g)
            mean = 0.2 * torch.randn(C, generator=g)
            vae = types.SimpleNamespace(std_of_means=std, mean_of_means=mean)
            ms = object.__new__(ref_pl.LTXMultiScalePipeline)
            ms.vae = vae
            t["per_channel_statistics.std-of-means"], t["per_channel_statistics.mean-of-means"] = std, mean
            t["upsample_latents"] = ms._upsample_latents(m, lat)
            t["adain"] = ref_pl.adain_filter_latent(latents=t["upsample_latents"], reference_latents=lat)
            t["adain_half"] = ref_pl.adain_filter_latent(t["upsample_latents"], lat, factor=0.5)
    save("g13_latent_upsampler", t, meta)


@torch.no_grad()
def g14():
    """retrieve_timesteps (schedule slicing for the second pass / strength < 1) and prepare_latents
    (given latents are re-noised to the first timestep), pipeline_ltx_video.py:125-198, 632-710."""
    print("G14 pipeline control")
    ref_shims.install_pipeline_leaves()
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl
    t, meta = {}, {"cases": []}
    shape = (1, 8, 3, 4, 6)

    def sch():
        return ref_rf.RectifiedFlowScheduler(num_train_timesteps=1000, shifting="SD3", base_resolution=None,
                                             target_shift_terminal=0.1)

    cases = [dict(num_inference_steps=10), dict(num_inference_steps=10, skip_initial_inference_steps=3),
             dict(num_inference_steps=10, skip_final_inference_steps=4),
             dict(num_inference_steps=30, skip_initial_inference_steps=17, skip_final_inference_steps=2),
             dict(num_inference_steps=10, max_timestep=0.7),
             dict(timesteps=[1.0, 0.9937, 0.9875, 0.9812, 0.975, 0.9094, 0.725, 0.4219]),
             dict(timesteps=[0.9094, 0.725, 0.4219], skip_initial_inference_steps=1)]
    for i, kw in enumerate(cases):
        s = sch()
        kw2 = dict(kw)
        n = kw2.pop("num_inference_steps", None)
        ts, cnt = ref_pl.retrieve_timesteps(s, n, "cpu", kw2.pop("timesteps", None), samples_shape=torch.Size(shape),
                                            **kw2)
        assert cnt == len(ts) and torch.equal(s.timesteps, ts)
        t[f"ts.{i}"] = ts
        meta["cases"].append(kw)
    pipe = object.__new__(ref_pl.LTXVideoPipeline)
    pipe.scheduler = sch()
    g = torch.Generator().manual_seed(90)
    lat = torch.randn(shape, generator=g)
    t["latents"] = lat
    draws = []

    def logged_randn(shp, generator=None, device=None, dtype=None, layout=None):
        n = ref_shims.randn_tensor(shp, generator=generator, device=device, dtype=dtype)
        draws.append(n)
        return n

    ref_pl.randn_tensor = logged_randn
    t["prepared"] = pipe.prepare_latents(lat, None, torch.tensor(0.725), shape, torch.float32, "cpu",
                                         torch.Generator().manual_seed(91))
    t["prepared_none"] = pipe.prepare_latents(None, None, torch.tensor(1.0), shape, torch.float32, "cpu",
                                              torch.Generator().manual_seed(92))
    t["noise.0"], t["noise.1"] = draws
    save("g14_pipeline_control", t, dict(meta, shape=list(shape), t0=0.725))


@torch.no_grad()
def g7():
    """G7 + G11 of SURVEY 8c: the reference's OWN ``LTXVideoPipeline.__call__`` (pipeline_ltx_video.py:762-1307) for
    config 1 -- 256x256x9, 2 denoise steps, fp32 on the CPU, CFG 3 + STG 1 (skip block 1, AttentionValues) + std-rescale 0.7
    -- on an instance made without __init__ (tiny reference DiT, the reference scheduler and patchifier).  Captured per
    step: what the transformer was fed and returned, the guided prediction handed to denoising_step, the latents after
    the step; plus the final unpatchified latents.  These pin the guidance block (:1183-1222), the per-step guidance
    tables (:959-1013) and the loop plumbing of the oracle.
    The one line of __call__ that cannot run without a GPU is ``negative_prompt_attention_mask.to("cuda")`` (:1041): the
    mask is handed in as a Tensor subclass whose ``.to("cuda")`` stays on the CPU.  Nothing of the reference is edited."""
    print("G7/G11 LTXVideoPipeline.__call__, config 1")
    import contextlib
    ref_shims.install_pipeline_leaves()
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl

    class _StaysOnCpu(torch.Tensor):
        def to(self, *a, **k):
            a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
            return torch.Tensor.to(self.as_subclass(torch.Tensor), *a, **k) if (a or k) else self.as_subclass(torch.Tensor)

    # head_dim 64 / widths that are multiples of 64, so that the PRODUCT can run the same weights (tests/test_gpu_model.py)
    cfg = dict(TINY_DIT, in_channels=128, out_channels=128, num_attention_heads=2, attention_head_dim=64,
               cross_attention_dim=128, caption_channels=128)
    model = build_dit(cfg, 95)
    # diffusers' ConfigMixin.__getattr__ forwards config keys as attributes (``self.transformer.in_channels``, :1266);
    # the shimmed ModelMixin does not, so the attribute is set here
    model.in_channels = cfg["in_channels"]
    draws = []

    def logged_randn(shp, generator=None, device=None, dtype=None, layout=None):
        n = ref_shims.randn_tensor(shp, generator=generator, device=device, dtype=dtype)
        draws.append(n)
        return n

    ref_pl.randn_tensor = logged_randn
    vae = ref_cva.CausalVideoAutoencoder.from_config(json.loads(json.dumps(jsonable(tiny_vae_b())))).eval()
    pipe = object.__new__(ref_pl.LTXVideoPipeline)
    pipe.transformer = model
    pipe.vae = vae
    pipe.patchifier = ref_sp.SymmetricPatchifier(patch_size=1)
    pipe.scheduler = ref_rf.RectifiedFlowScheduler(sampler="Uniform", shifting="SD3", base_resolution=None,
                                                   target_shift_terminal=0.1)
    pipe.vae_scale_factor, pipe.video_scale_factor = 32, 8
    pipe.allowed_inference_steps = None
    pipe._execution_device = torch.device("cpu")

    class _Bar:
        def update(self, *a):
            pass

    pipe.progress_bar = lambda total=None: contextlib.nullcontext(_Bar())

    rec = {}
    calls = {"n": 0}
    fwd = model.forward

    def rec_forward(hidden_states, **kw):
        i = calls["n"]
        out = fwd(hidden_states, **kw)
        rec[f"model_in.{i}"] = hidden_states.float()
        rec[f"timestep.{i}"] = kw["timestep"].float()
        rec[f"raw_pred.{i}"] = out[0].float()
        if kw.get("skip_layer_mask") is not None:
            rec[f"skip_layer_mask.{i}"] = kw["skip_layer_mask"].float()
        calls["n"] += 1
        return out

    model.forward = rec_forward
    den = ref_pl.LTXVideoPipeline.denoising_step
    steps_seen = {"n": 0}

    def rec_denoise(latents, noise_pred, current_timestep, conditioning_mask, t, extra_step_kwargs, **kw):
        i = steps_seen["n"]
        out = den(pipe, latents, noise_pred, current_timestep, conditioning_mask, t, extra_step_kwargs, **kw)
        rec[f"guided.{i}"] = noise_pred.float()
        rec[f"latents_in.{i}"] = latents.float()
        rec[f"latents_out.{i}"] = out.float()
        steps_seen["n"] += 1
        return out

    pipe.denoising_step = rec_denoise

    g = torch.Generator().manual_seed(96)
    T, cap = 12, cfg["caption_channels"]
    pos, neg = torch.randn(1, T, cap, generator=g), torch.randn(1, T, cap, generator=g)
    pmask, nmask = torch.ones(1, T), torch.ones(1, T)
    pmask[:, 8:] = 0
    nmask[:, 3:] = 0
    kw = dict(height=256, width=256, num_frames=9, frame_rate=25.0, num_inference_steps=2, guidance_scale=3.0,
              stg_scale=1.0, rescaling_scale=0.7, skip_block_list=[1], skip_layer_strategy=SkipLayerStrategy.AttentionValues)
    out = pipe(prompt_embeds=pos, prompt_attention_mask=pmask, negative_prompt_embeds=neg,
               negative_prompt_attention_mask=nmask.as_subclass(_StaysOnCpu), generator=torch.Generator().manual_seed(97),
               output_type="latent", return_dict=False, is_video=True, vae_per_channel_normalize=True, joint_pass=True,
               ltxv_model=types.SimpleNamespace(_interrupt=False), **kw)[0]
    assert calls["n"] == 2 and steps_seen["n"] == 2 and len(draws) == 1
    t = dict(rec)
    t.update({f"w.{k}": v for k, v in model.state_dict().items()})
    t["prompt_embeds"], t["negative_prompt_embeds"] = pos, neg
    t["prompt_attention_mask"], t["negative_prompt_attention_mask"] = pmask, nmask
    t["noise"] = draws[0]
    t["timesteps"] = pipe.scheduler.timesteps.float()
    t["out_latents"] = out.float()

    # a second run with per-step guidance tables (the 13B multi-scale settings' form): lists + guidance_timesteps
    calls["n"] = steps_seen["n"] = 0
    rec.clear()
    draws.clear()
    kw2 = dict(kw, num_inference_steps=3, guidance_scale=[1.0, 3.0, 2.0], stg_scale=[0.0, 1.0, 0.5],
               rescaling_scale=[1.0, 0.7, 0.9], guidance_timesteps=[1.0, 0.9, 0.3], skip_block_list=[[], [1], [0]])
    out2 = pipe(prompt_embeds=pos, prompt_attention_mask=pmask, negative_prompt_embeds=neg,
                negative_prompt_attention_mask=nmask.as_subclass(_StaysOnCpu), generator=torch.Generator().manual_seed(98),
                output_type="latent", return_dict=False, is_video=True, vae_per_channel_normalize=True, joint_pass=True,
                ltxv_model=types.SimpleNamespace(_interrupt=False), **kw2)[0]
    assert calls["n"] == 3
    t.update({f"tables.{k}": v for k, v in rec.items()})
    t["tables.noise"] = draws[0]
    t["tables.timesteps"] = pipe.scheduler.timesteps.float()
    t["tables.out_latents"] = out2.float()
    save("g7_pipeline_call", t, dict(cfg=cfg, kwargs={k: (v if not isinstance(v, SkipLayerStrategy) else v.name) for k, v in kw.items()},
                                     tables_kwargs={k: (v if not isinstance(v, SkipLayerStrategy) else v.name) for k, v in kw2.items()},
                                     T=T, grid=[2, 8, 8]))



@torch.no_grad()
def g10c():
    """The decoder block kinds the demo / OURS configs do not use (causal_video_autoencoder.py:671-684, 1183-1195):
    compress_time (2,1,1), compress_space (1,2,2) and inject_noise ResnetBlock3Ds, through the reference's own
    Decoder.forward.  The noise the reference draws inside _feed_spatial_noise (unseeded torch.randn of shape [H, W]) is
    made reproducible with torch.manual_seed and recorded by drawing the same sequence again."""
    print("G10c decoder block variants")
    cfg = {"_class_name": "CausalVideoAutoencoder", "dims": 3, "in_channels": 3, "out_channels": 3, "latent_channels": 8,
           "encoder_blocks": [["res_x", {"num_layers": 1}]],
           "decoder_blocks": [["res_x", {"num_layers": 1}], ["compress_space", {}], ["res_x", {"num_layers": 2, "inject_noise": True}],
                              ["compress_time", {}], ["res_x", {"num_layers": 1, "inject_noise": True}]],
           "scaling_factor": 1.0, "norm_layer": "pixel_norm", "patch_size": 2, "latent_log_var": "uniform",
           "use_quant_conv": False, "causal_decoder": False, "timestep_conditioning": True,
           "spatial_padding_mode": "replicate", "encoder_base_channels": 8, "decoder_base_channels": 8}
    torch.manual_seed(120)
    vae = ref_cva.CausalVideoAutoencoder.from_config(jsonable(cfg)).eval()
    g = torch.Generator().manual_seed(121)
    for n, prm in vae.decoder.named_parameters():
        if "per_channel_scale" in n:                      # zero-initialised in the reference: make the path visible
            prm.copy_(0.3 * torch.randn(prm.shape, generator=g))
    t = {"sd." + k: v for k, v in vae.state_dict().items() if k.startswith("decoder.")}
    z = torch.randn(1, 8, 3, 4, 5, generator=g)
    ts = torch.tensor([0.05])
    t["z"], t["timestep"] = z, ts
    # reversed order of decoder_blocks: res_x(noise, 1 layer) @ (3,4,5), compress_time -> (5,4,5), res_x(noise, 2 layers),
    # compress_space -> (5,8,10), res_x
    torch.manual_seed(122)
    out = vae.decoder(z, target_shape=(1, 3, 5, 16, 20), timestep=ts)
    torch.manual_seed(122)
    shapes = [(4, 5)] * 2 + [(4, 5)] * 4
    for i, shp in enumerate(shapes):
        t[f"noise.{i}"] = torch.randn(shp)
    t["out"] = out
    save("g10c_decoder_variants", t, dict(cfg=cfg, noise_shapes=[list(s) for s in shapes]))


@torch.no_grad()
def g0():
    """G0: the reference's own ``get_timestep_embedding`` (ltx_video/models/transformers/embeddings.py:10-50) on the
    timestep values the path feeds it (scaled by timestep_scale_multiplier = 1000, fractional, zero, per-frame lists) in
    the two parameterisations used: (256, flip, shift 0) of PixArtAlphaCombinedTimestepSizeEmbeddings, and the function's
    defaults.  Pins oracle/leaves.py::get_timestep_embedding (CPU) and the ltxmi_timestep_embedding_bf16 kernel (GPU)."""
    print("G0 sinusoidal timestep embedding")
    from ltx_video.models.transformers.embeddings import get_timestep_embedding
    g = torch.Generator().manual_seed(5)
    ts = torch.cat([torch.tensor([0.0, 1.0, 1000.0, 999.0, 700.0, 50.0, 0.5, 123.456]),
                    1000.0 * torch.rand(24, generator=g)])
    t = {"timesteps": ts,
         "emb_256_flip_shift0": get_timestep_embedding(ts, 256, flip_sin_to_cos=True, downscale_freq_shift=0.0),
         "emb_64_defaults": get_timestep_embedding(ts, 64),
         "emb_33_odd_scale2": get_timestep_embedding(ts, 33, flip_sin_to_cos=False, downscale_freq_shift=1, scale=2.0,
                                                     max_period=1000)}
    save("g0_timestep_embedding", t, {})


@torch.no_grad()
def g15():
    """G15: the reference's own ``LTXMultiScalePipeline.__call__`` (pipeline_ltx_video.py:1741-1905) called with EXACTLY the
    keyword arguments of ``LTXV.generate`` (ltxv.py:420-445): the YAML dict spread into the call (every key of
    configs/ltxv-13b-0.9.7-dev.yaml except ``stg_mode``, which ltxv.py:405 deletes; the per-step tables shortened to the
    tiny model's 2 blocks and a handful of steps), string prompts, ``output_type="pt"``, ``VAE_tile_size``, ``device``,
    ``callback`` -- fp32 on the CPU, on instances made without the diffusers base-class __init__.  The T5 pair is the
    test double of tests/fake_t5.py (the text encoder is outside the path; the same double drives the product in the
    tests).  DiT = G7's (same config and seed: its weights are in g7_pipeline_call); VAE decoder and latent upsampler
    weights come from the oracle's seeded initialisers (too large to commit: regenerated in the tests, fingerprints in
    the manifest).  Recorded: every noise draw, the pass-1 latents, the upsampled + AdaIN latents, the callback trace and
    the final video."""
    print("G15 LTXMultiScalePipeline.__call__ with ltxv.py's keyword arguments")
    import contextlib
    import tempfile
    ref_shims.install_pipeline_leaves()
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl
    import ltx_video.models.autoencoders.latent_upsampler as ref_lu
    from oracle import upsampler as ou, vae as ov
    from tests.fake_t5 import FakeTextEncoder, FakeTokenizer

    class _StaysOnCpu(torch.Tensor):
        def to(self, *a, **k):
            a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
            return torch.Tensor.to(self.as_subclass(torch.Tensor), *a, **k) if (a or k) else self.as_subclass(torch.Tensor)

    cfg = dict(TINY_DIT, in_channels=128, out_channels=128, num_attention_heads=2, attention_head_dim=64,
               cross_attention_dim=128, caption_channels=128)
    model = build_dit(cfg, 95)                                  # = G7's model
    model.in_channels = cfg["in_channels"]
    # the reference's demo config (0.9.5+-style decoder); the encoder (unused: no conditioning media) stays at 8 base
    # channels, but its block list must be the real one: the pipeline derives its (8, 32, 32) scale factors from it
    vcfg = jsonable(dict(ref_cva.create_video_autoencoder_demo_config(latent_channels=128), decoder_base_channels=64,
                         encoder_base_channels=8))
    vsd = ov.init_state_dict(vcfg, seed=150)
    std, mean = vsd["per_channel_statistics.std-of-means"], vsd["per_channel_statistics.mean-of-means"]
    vae = ref_cva.CausalVideoAutoencoder.from_config(jsonable(vcfg)).eval()
    missing, unexpected = vae.decoder.load_state_dict({k[len("decoder."):]: v for k, v in vsd.items()
                                                       if k.startswith("decoder.")}, strict=False)
    assert not unexpected and all("timestep_scale_multiplier" in k or "last_" in k for k in missing), (missing, unexpected)
    vae.register_buffer("std_of_means", std)
    vae.register_buffer("mean_of_means", mean)
    ucfg = dict(in_channels=128, mid_channels=64, num_blocks_per_stage=1, dims=3, spatial_upsample=True,
                temporal_upsample=False)
    usd = ou.init_state_dict(ucfg, seed=152)
    ups = ref_lu.LatentUpsampler.from_config(ucfg).eval()
    ups.load_state_dict(usd)

    draws = []

    def logged_randn(shp, generator=None, device=None, dtype=None, layout=None):
        n = ref_shims.randn_tensor(shp, generator=generator, device=device, dtype=dtype)
        draws.append(n)
        return n

    ref_pl.randn_tensor = logged_randn
    pipe = object.__new__(ref_pl.LTXVideoPipeline)
    pipe.transformer, pipe.vae = model, vae
    pipe.patchifier = ref_sp.SymmetricPatchifier(patch_size=1)
    pipe.scheduler = ref_rf.RectifiedFlowScheduler(sampler="Uniform", shifting="SD3", base_resolution=None,
                                                   target_shift_terminal=0.1)
    pipe.tokenizer, pipe.text_encoder = FakeTokenizer(), FakeTextEncoder(cfg["caption_channels"], seed=153).eval()
    pipe.vae_scale_factor, pipe.video_scale_factor = 32, 8
    pipe.allowed_inference_steps = None
    pipe._execution_device = torch.device("cpu")
    pipe.image_processor = ref_shims.VaeImageProcessor(vae_scale_factor=32)

    class _Bar:
        def update(self, *a):
            pass

    pipe.progress_bar = lambda total=None: contextlib.nullcontext(_Bar())
    enc = pipe.encode_prompt

    def encode_keeping_the_mask_on_the_cpu(*a, **k):            # __call__ :1041 does negative_prompt_attention_mask.to("cuda")
        pe, pm, ne, nm = enc(*a, **k)
        rec["prompt_embeds"], rec["prompt_attention_mask"] = pe.float(), pm.float()
        rec["negative_prompt_embeds"], rec["negative_prompt_attention_mask"] = ne.float(), nm.float()
        return pe, pm, ne, nm.as_subclass(_StaysOnCpu)

    pipe.encode_prompt = encode_keeping_the_mask_on_the_cpu
    ms = ref_pl.LTXMultiScalePipeline(pipe, ups)
    rec = {}
    up = ms._upsample_latents

    def rec_upsample(upsampler, latents):
        out = up(upsampler, latents)
        rec["pass1_latents"], rec["upsampled"] = latents.float(), out.float()
        return out

    ms._upsample_latents = rec_upsample
    trace = []

    def callback(i, preview, start, **kw):
        trace.append([int(i), None if preview is None else list(preview.shape), bool(start), int(kw.get("pass_no", 0)),
                      kw.get("override_num_inference_steps")])
        if preview is not None:                                 # the latents after step i of that pass, (c, f, h, w)
            rec[f"preview.{kw.get('pass_no', 0)}.{i}"] = preview.float().clone()

    # ltxv.py:309-312: pipeline_config = yaml.safe_load(the dev YAML); :404-405: stg_mode read and deleted.  Same keys; the
    # guidance tables keep their form (lists over guidance_timesteps, per-entry skip lists) at the tiny model's size.
    pipeline_config = {
        "pipeline_type": "multi-scale", "checkpoint_path": "ltxv-13b-0.9.7-dev.safetensors", "downscale_factor": 0.6666666,
        "spatial_upscaler_model_path": "ltxv-spatial-upscaler-0.9.7.safetensors", "decode_timestep": 0.05,
        "decode_noise_scale": 0.025, "text_encoder_model_name_or_path": "PixArt-alpha/PixArt-XL-2-1024-MS",
        "precision": "bfloat16", "sampler": "from_checkpoint", "prompt_enhancement_words_threshold": 120,
        "prompt_enhancer_image_caption_model_name_or_path": "MiaoshouAI/Florence-2-large-PromptGen-v2.0",
        "prompt_enhancer_llm_model_name_or_path": "unsloth/Llama-3.2-3B-Instruct", "stochastic_sampling": False,
        "first_pass": {"guidance_scale": [1, 1, 6, 8, 6, 1, 1], "stg_scale": [0, 0, 4, 4, 4, 2, 1],
                       "rescaling_scale": [1, 1, 0.5, 0.5, 1, 1, 1],
                       "guidance_timesteps": [1.0, 0.9, 0.75, 0.6, 0.4, 0.2, 0.1],      # (spread over the 5-step schedule)
                       "skip_block_list": [[], [1], [0, 1], [1], [1], [0], [1]], "num_inference_steps": 30,
                       "skip_final_inference_steps": 1, "cfg_star_rescale": True},
        "second_pass": {"guidance_scale": [1], "stg_scale": [1], "rescaling_scale": [1], "guidance_timesteps": [1.0],
                        "skip_block_list": [1], "num_inference_steps": 30, "skip_initial_inference_steps": 3,
                        "cfg_star_rescale": True},
    }
    call = dict(num_inference_steps1=6, num_inference_steps2=5, output_type="pt", callback_on_step_end=None,
                height=96, width=192, num_frames=9, frame_rate=30,
                prompt="a red fox runs through fresh snow at dawn", prompt_attention_mask=None,
                negative_prompt="worst quality, blurry, jittery", negative_prompt_attention_mask=None,
                media_items=None, strength=1.0, conditioning_items=None, is_video=True, vae_per_channel_normalize=True,
                image_cond_noise_scale=0.15, mixed_precision=False, VAE_tile_size=(0, 0), device="cpu")
    holder = types.SimpleNamespace(_interrupt=False)
    like_draws, plain_randn_like = [], torch.randn_like

    def logged_randn_like(x, *a, **k):                         # recorded as drawn (other code shares the global RNG)
        n = plain_randn_like(x, *a, **k)
        like_draws.append(n.clone())
        return n

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:                 # __call__ :1290 writes "lala.pt" into the working directory
        os.chdir(tmp)
        try:
            torch.manual_seed(154)                             # the decode noise is torch.randn_like: the global RNG (:1271)
            torch.randn_like = logged_randn_like
            images = ms(**json.loads(json.dumps(pipeline_config)), ltxv_model=holder,
                        skip_layer_strategy=SkipLayerStrategy.AttentionValues,
                        generator=torch.Generator().manual_seed(155), callback=callback, **call)
        finally:
            torch.randn_like = plain_randn_like
            os.chdir(cwd)
    assert len(draws) == 2 and len(like_draws) == 1, [tuple(d.shape) for d in draws + like_draws]
    t = dict(rec)
    t["decode_noise"] = like_draws[0]
    t["noise.0"], t["noise.1"] = draws
    t["images"] = images.float()

    def fingerprint(sd):
        return float(sum(v.double().abs().sum() for v in sd.values()))

    save("g15_multiscale_call", t, dict(
        dit_cfg=cfg, dit_weights="g7_pipeline_call (w.*)", vae_cfg=jsonable(vcfg), vae_seed=150, upsampler_cfg=ucfg,
        upsampler_seed=152, text_encoder_seed=153, vae_fingerprint=fingerprint(vsd), upsampler_fingerprint=fingerprint(usd),
        pipeline_config=pipeline_config, call=call, skip_layer_strategy="AttentionValues", callback_trace=trace,
        images_shape=list(images.shape)))


def _sig_record(fn):
    """names, kinds and defaults (repr; enum members by name) of a callable's parameters -- data about the reference's
    interface, nothing of its code."""
    import enum
    import inspect
    out = []
    for prm in inspect.signature(fn).parameters.values():
        d = None
        if prm.default is not inspect.Parameter.empty:
            v = prm.default
            d = f"{type(v).__name__}.{v.name}" if isinstance(v, enum.Enum) else repr(v)
        out.append({"name": prm.name, "kind": prm.kind.name, "has_default": prm.default is not inspect.Parameter.empty,
                    "default": d})
    return out


def _sig_record_ast(path, func):
    """The same record for a module that cannot be imported here (wan/distributed needs xfuser): parsed from the source
    text with ``ast`` -- only the parameter list is read."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read())
    node = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == func)
    a = node.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    out = []
    for arg, d in zip(pos, defaults):
        out.append({"name": arg.arg, "kind": "POSITIONAL_OR_KEYWORD", "has_default": d is not None,
                    "default": None if d is None else ast.unparse(d)})
    if a.vararg:
        out.append({"name": a.vararg.arg, "kind": "VAR_POSITIONAL", "has_default": False, "default": None})
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": arg.arg, "kind": "KEYWORD_ONLY", "has_default": d is not None,
                    "default": None if d is None else ast.unparse(d)})
    if a.kwarg:
        out.append({"name": a.kwarg.arg, "kind": "VAR_KEYWORD", "has_default": False, "default": None})
    return out


def signatures():
    """``inspect.signature`` of every public entry point of the hot path, taken from the reference's own
    objects, into tests/golden/signatures.json.  tests/test_host_logic.py compares the product's against it (same
    names, order, kinds and defaults; only trailing keyword extras allowed)."""
    print("signatures of the reference's entry points")
    ref_shims.install_pipeline_leaves()
    import ltx_video.pipelines.pipeline_ltx_video as ref_pl
    import ltx_video.models.autoencoders.vae as ref_vae
    import ltx_video.models.autoencoders.latent_upsampler as ref_up
    import wan.modules.attention as ref_wa
    pl, ms = ref_pl.LTXVideoPipeline, ref_pl.LTXMultiScalePipeline
    table = {
        "Transformer3DModel.__init__": ref_t3.Transformer3DModel.__init__,
        "Transformer3DModel.forward": ref_t3.Transformer3DModel.forward,
        "Transformer3DModel.precompute_freqs_cis": ref_t3.Transformer3DModel.precompute_freqs_cis,
        "Transformer3DModel.create_skip_layer_mask": ref_t3.Transformer3DModel.create_skip_layer_mask,
        "BasicTransformerBlock.forward": ref_attn.BasicTransformerBlock.forward,
        "Attention.set_processor": ref_attn.Attention.set_processor,
        "Attention.forward": ref_attn.Attention.forward,
        "AttnProcessor2_0.__call__": ref_attn.AttnProcessor2_0.__call__,
        "pay_attention": ref_wa.pay_attention,
        "CausalVideoAutoencoder.decode": ref_cva.CausalVideoAutoencoder.decode,
        "CausalVideoAutoencoder.encode": ref_cva.CausalVideoAutoencoder.encode,
        "CausalVideoAutoencoder.enable_z_tiling": ref_cva.CausalVideoAutoencoder.enable_z_tiling,
        "CausalVideoAutoencoder.set_tiling_params": ref_cva.CausalVideoAutoencoder.set_tiling_params,
        "CausalVideoAutoencoder.get_VAE_tile_size": ref_cva.CausalVideoAutoencoder.get_VAE_tile_size,
        "Decoder.forward": ref_cva.Decoder.forward,
        "CausalConv3d.forward": ref_cc3.CausalConv3d.forward,
        "vae_decode": ref_ve.vae_decode,
        "vae_encode": ref_ve.vae_encode,
        "un_normalize_latents": ref_ve.un_normalize_latents,
        "normalize_latents": ref_ve.normalize_latents,
        "latent_to_pixel_coords": ref_ve.latent_to_pixel_coords,
        "SymmetricPatchifier.patchify": ref_sp.SymmetricPatchifier.patchify,
        "SymmetricPatchifier.unpatchify": ref_sp.SymmetricPatchifier.unpatchify,
        "RectifiedFlowScheduler.__init__": ref_rf.RectifiedFlowScheduler.__init__,
        "RectifiedFlowScheduler.set_timesteps": ref_rf.RectifiedFlowScheduler.set_timesteps,
        "RectifiedFlowScheduler.step": ref_rf.RectifiedFlowScheduler.step,
        "RectifiedFlowScheduler.add_noise": ref_rf.RectifiedFlowScheduler.add_noise,
        "RectifiedFlowScheduler.scale_model_input": ref_rf.RectifiedFlowScheduler.scale_model_input,
        "LTXVideoPipeline.__init__": pl.__init__,
        "LTXVideoPipeline.__call__": pl.__call__,
        "LTXVideoPipeline.encode_prompt": pl.encode_prompt,
        "LTXVideoPipeline.check_inputs": pl.check_inputs,
        "LTXVideoPipeline.prepare_latents": pl.prepare_latents,
        "LTXVideoPipeline.prepare_conditioning": pl.prepare_conditioning,
        "LTXVideoPipeline.resize_tensor": pl.resize_tensor,
        "LTXMultiScalePipeline.__init__": ms.__init__,
        "LTXMultiScalePipeline.__call__": ms.__call__,
        "LTXMultiScalePipeline._upsample_latents": ms._upsample_latents,
        "retrieve_timesteps": ref_pl.retrieve_timesteps,
        "adain_filter_latent": ref_pl.adain_filter_latent,
        "ConditioningItem": ref_pl.ConditioningItem,
        "LatentUpsampler.__init__": ref_up.LatentUpsampler.__init__,
        "LatentUpsampler.forward": ref_up.LatentUpsampler.forward,
    }
    rec = {k: _sig_record(v) for k, v in table.items()}
    usp = ref_shims.REFERENCE_ROOT + "/wan/distributed/xdit_context_parallel.py"
    rec["wan.usp_dit_forward"] = _sig_record_ast(usp, "usp_dit_forward")
    rec["wan.usp_attn_forward"] = _sig_record_ast(usp, "usp_attn_forward")
    rec["wan.shard_model"] = _sig_record_ast(ref_shims.REFERENCE_ROOT + "/wan/distributed/fsdp.py", "shard_model")
    with open(os.path.join(OUT, "signatures.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(f"  signatures.json: {len(rec)} entry points")


CASES = {"signatures": signatures, "g0": g0, "g15": g15, "g10c": g10c, "g7": g7, "g14": g14, "g13": g13, "g1": g1_g2, "g3": g3_g4_g5, "g6": g6, "g8": g8_g9, "g10": g10, "g11": g11, "g12": g12}


def main():
    """No arguments: regenerate everything.  With case names (``g11`` ...): only those, merged
    into the existing manifest."""
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    mpath = os.path.join(OUT, "manifest.json")
    if sys.argv[1:] and os.path.exists(mpath):
        with open(mpath) as f:
            MANIFEST.update(json.load(f))
    for n in names:
        CASES[n]()
    with open(mpath, "w") as f:
        json.dump(MANIFEST, f, indent=1, default=list)
    print("done")


if __name__ == "__main__":
    main()
