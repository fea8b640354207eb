#!/usr/bin/env python3
"""bench.py -- denoise-steps/s (+ VAE-decode frames/s) for LTX-Video 2B t2v 768x512x97 on MI355X.

A "step" is ONE denoise step of BASELINE.json configs[1] on synthetic latents already resident
in HBM: Transformer3DModel.forward at B_eff = 3 (uncond + text + STG-perturbed rows, as the 2B dev
YAML enables: guidance 3, stg 1, skip block 19), N = 13*16*24 = 4992 tokens, T = 256 text tokens,
28 layers, bf16, random-init weights of the 2B architecture, followed by the fused
CFG-star/STG/rescale + Euler update.  `value` = steps/s over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU.  `value` = replicas (every rank denoises its own video, RCCL only for the
barrier / max-reduce of the clock; "scaling": "weak").  The same run then times the sequence-sharded mode the
north star names (ONE video, tokens sharded over the ranks, Ulysses all-to-all inside self-attention,
ltxmi/distributed.py) and reports it in the same line under "ulysses" ("scaling": "strong").
The JSON line also carries:
  roofline      the dominant kernel of this workload = the self-attention kernel (28 launches per step, the
                largest single share of a step; MFMA-bound): algorithmic FLOP per launch / average launch time
                from HIP events recorded on the launch stream inside the timed region
  roofline_gemm the same for the FF up-projection GEMM (the largest GEMM instance)
  step_ms       median / p10 / p90 of the timed steps (HIP events per step)
  b_eff_1       the same step with ONE cond (no CFG / STG rows)
  attention     the self-attention kernel at the north-star stress shape (N = 98304, 1 layer, B = 1), at config 3's
                N = 13376 and at config 4's Wan shape [1, 32760, 12, 128] (+ cross-attention, 512 text keys)
  vae_decode    frames/s of CausalVideoAutoencoder.decode for the same video (z [1,128,13,16,24]) and for
                config 5 (z [1,128,33,23,40] -> 257 frames of 1280 x 720, z-tiled by 4 latent frames)
  cpu_baseline  the CPU oracle (oracle/dit.py, fp32, host cores): ONE full 28-layer forward for one cond, measured directly
  cpu_baseline_vae  the CPU oracle's VAE decode (oracle/vae.py): config 1 whole, a slab of config 2 (extrapolated by FLOP)
  vae_decode.roofline_conv  the dominant convolution launch against the MFMA roof and the HBM roof
Every extra leg checks its output (finite + a band of rows / a second implementation) before it reports a time.
"""
import argparse
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
D, H, DH, L, FF, T_TEXT, C_LAT = 2048, 32, 64, 28, 8192, 256, 128
GRID = (13, 16, 24)                  # 768x512x97 -> latent frames/height/width
N_TOK = GRID[0] * GRID[1] * GRID[2]  # 4992
GRID_CONFIG3 = (16, 22, 38)          # 1216x704x121 -> 13376 tokens (BASELINE config 3, i2v)
NUM_CONDS = 3


def fractional_coords(device, grid=None):
    """What LTXVideoPipeline hands to precompute_freqs_cis (pipeline_ltx_video.py:1086-1088): seconds on the time
    axis (25 fps, causal fix), pixels on y / x.  Product-side helpers only: nothing under oracle/ is used outside
    the cpu_baseline leg."""
    from ltxmi.patchifier import SymmetricPatchifier, latent_to_pixel_coords_from_factors
    coords = SymmetricPatchifier(1).get_latent_coords(*(grid or GRID), 1, device)
    pc = latent_to_pixel_coords_from_factors(coords, (8, 32, 32), causal_fix=True).to(torch.float32)
    pc[:, 0] = pc[:, 0] * (1.0 / 25.0)
    return pc


def build_transformer(device, layers=L, seed=0):
    import ltxmi
    from ltxmi.loading import NATIVE_2B_TRANSFORMER_CONFIG       # OURS_TRANSFORMER_CONFIG, diffusers_config_mapping.py:74-105
    keep = ("num_attention_heads", "attention_head_dim", "in_channels", "out_channels", "num_layers", "cross_attention_dim",
            "caption_channels", "attention_bias", "activation_fn", "norm_elementwise_affine", "norm_eps", "qk_norm",
            "standardization_norm", "positional_embedding_type", "positional_embedding_theta",
            "positional_embedding_max_pos", "timestep_scale_multiplier")
    cfg = {k: NATIVE_2B_TRANSFORMER_CONFIG[k] for k in keep}
    cfg.update(adaptive_norm="single_scale_shift", num_layers=layers)
    torch.manual_seed(seed)
    with torch.device(device):
        m = ltxmi.Transformer3DModel(**cfg)
    for p in m.parameters():                       # module default init, bf16
        p.requires_grad_(False)
    return m.to(dtype=torch.bfloat16).eval(), cfg


def synth_inputs(device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    emb = torch.randn(1, T_TEXT, 4096, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    neg = torch.randn(1, T_TEXT, 4096, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    mask = torch.zeros(1, T_TEXT, device=device)
    mask[:, :96] = 1                               # first 96 tokens valid
    nmask = torch.zeros(1, T_TEXT, device=device)
    nmask[:, :8] = 1
    return emb, mask, neg, nmask


class StepRunner:
    """The loop body of LTXVideoPipeline.__call__ at a fixed timestep schedule position."""

    def __init__(self, device):
        import ltxmi
        from ltxmi import ops
        from ltxmi.scheduler import RectifiedFlowScheduler
        self.ops = ops
        self.ltxmi = ltxmi
        self.m, self.cfg = build_transformer(device)
        emb, mask, neg, nmask = synth_inputs(device)
        self.embeds = torch.cat([neg, emb, emb])
        self.mask = torch.cat([nmask, mask, mask])
        self.device = device
        self.set_grid(GRID)
        self.skip = self.m.create_skip_layer_mask(1, NUM_CONDS, NUM_CONDS - 1, [19])
        self.ws = torch.empty(ops.GUIDANCE_WORKSPACE_FLOATS, device=device)
        sch = RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1)       # OURS_SCHEDULER_CONFIG
        sch.set_timesteps(40, samples_shape=(1, C_LAT) + GRID, device="cpu")
        ts = sch.timesteps
        self.t = float(ts[10])
        self.dt = float(ts[10] - ts[11])
        self.t_dev = torch.full((NUM_CONDS, 1), self.t, device=device)
        self.t_dev1 = torch.full((1, 1), self.t, device=device)
        self.t_global = self.t_dev

        class Holder:
            _interrupt = False
        self.holder = Holder()

    def set_grid(self, grid, image_conditioned=False):
        """Latents, RoPE tables (and, image_conditioned: the per-token timesteps of an i2v step -- the first latent frame
        is the conditioning image, pipeline_ltx_video.py:1157-1166) for a latent grid (frames, height, width)."""
        device = self.device
        self.grid = tuple(grid)
        self.n_tok = grid[0] * grid[1] * grid[2]
        g = torch.Generator(device=device).manual_seed(1)
        self.latents = torch.randn(1, self.n_tok, C_LAT, generator=g, device=device, dtype=torch.float32)
        self.freqs = self.m.precompute_freqs_cis(fractional_coords(device, self.grid))
        if image_conditioned:
            t = torch.full((NUM_CONDS, self.n_tok), self.t, device=device)
            t[:, : grid[1] * grid[2]] = 0.0
            self.t_dev = t
        elif hasattr(self, "t_global"):
            self.t_dev = self.t_global

    def enable_ulysses(self):
        """One video over all ranks: tokens sharded, all-to-all inside self-attention (ltxmi.distributed)."""
        from ltxmi import distributed as sp
        sp.enable_sequence_parallel(self.m)
        self.sp = sp

    @torch.no_grad()
    def step_one_cond(self):
        """B_eff = 1: the text row only (no CFG, no STG; guidance scale 1 leaves the prediction as it is)."""
        x = self.latents.to(torch.bfloat16)
        noise_pred = self.m(x, freqs_cis=self.freqs, encoder_hidden_states=self.embeds[1:2],
                            encoder_attention_mask=self.mask[1:2], timestep=self.t_dev1, latent_shape=self.grid,
                            ltxv_model=self.holder, return_dict=False)[0]
        self.ops.guidance_step_(noise_pred, self.latents, self.dt, 1.0, 0.0, 1.0, False, False, False, self.ws)

    @torch.no_grad()
    def step(self, stg_alias_blocks=0):
        x = self.latents.to(torch.bfloat16).expand(NUM_CONDS, -1, -1)
        if getattr(self, "sp", None) is not None:
            noise_pred = self.sp.usp_dit_forward(
                self.m, x, self.freqs, encoder_hidden_states=self.embeds, encoder_attention_mask=self.mask,
                timestep=self.t_dev, skip_layer_mask=self.skip,
                skip_layer_strategy=self.ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=self.grid,
                ltxv_model=self.holder)[0]
            self.ops.guidance_step_(noise_pred.contiguous(), self.latents, self.dt, 3.0, 1.0, 0.7, True, True, True,
                                    self.ws)
            return
        noise_pred = self.m(x, freqs_cis=self.freqs, encoder_hidden_states=self.embeds,
                            encoder_attention_mask=self.mask, timestep=self.t_dev, skip_layer_mask=self.skip,
                            skip_layer_strategy=self.ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=self.grid,
                            ltxv_model=self.holder, return_dict=False, stg_alias_blocks=stg_alias_blocks)[0]
        self.ops.guidance_step_(noise_pred, self.latents, self.dt, 3.0, 1.0, 0.7, True, True, True, self.ws)


def time_config3_step(runner, ops, steps):
    """BASELINE config 3 on ONE GPU (the N = 1 anchor of its 1 -> 8 curve): 2B i2v 1216x704x121 = 16 x 22 x 38 = 13 376 tokens,
    first latent frame conditioned (per-token timesteps, pipeline_ltx_video.py:1123-1166), B_eff 3, 28 layers + guidance /
    Euler.  Parity at this size: tests/test_gpu_model.py::test_transformer_config3_*."""
    n3 = GRID_CONFIG3[0] * GRID_CONFIG3[1] * GRID_CONFIG3[2]
    steps = max(3, min(steps, 10))
    runner.set_grid(GRID_CONFIG3, image_conditioned=True)
    try:
        for _ in range(2):
            runner.step()
        key = ("attention", NUM_CONDS, H, n3, n3, DH)
        ops.watch_launches([key])
        e, per = timed_steps(runner.step, steps, None)
        at = ops.launch_times_ms().get(key, [])
        ops.watch_launches(None)
        assert torch.isfinite(runner.latents).all(), "config 3: non-finite latents"
    finally:
        runner.set_grid(GRID)
    # per cond: GEMM 28 D^2 N L (+ text K/V) = 44.0, attention 4 N^2 D L = 41.1 TFLOP
    tflop = NUM_CONDS * (28.0 * D * D * n3 * L + 4.0 * D * D * T_TEXT * L + 4.0 * n3 * n3 * D * L) / 1e12
    ms = e / steps * 1e3
    at_ms = sum(at) / max(len(at), 1)
    at_tf = 4.0 * NUM_CONDS * n3 * n3 * D / (at_ms * 1e-3) / 1e12 if at_ms > 0 else 0.0
    return {"workload": "LTX-Video 2B i2v 1216x704x121 (13376 tokens, first latent frame conditioned: per-token timesteps), "
                        "B_eff 3, 28 layers + guidance/Euler, ONE GPU", "tokens": n3, "steps": steps,
            "denoise_steps_per_s": round(steps / e, 4), "ms_per_step": round(ms, 2),
            "step_ms": {"median": round(pct(per, 0.5), 3), "p10": round(pct(per, 0.1), 3), "p90": round(pct(per, 0.9), 3)},
            "algorithmic_tflop_per_step": round(tflop, 1), "step_tflops": round(tflop / (ms * 1e-3), 1),
            "attention_launch_ms": round(at_ms, 4), "attention_tflops": round(at_tf, 1),
            "attention_frac_of_mfma_peak": round(at_tf / MFMA_BF16_PEAK_TFLOPS, 4)}


def ulysses_projection(runner, ops, steps):
    """NOT a measurement of sequence parallelism -- a compute-only projection from ONE GPU, for rounds without a
    multi-GPU node: one rank's share of a P-rank Ulysses step (N/P local tokens through every token-local kernel, the
    destination-major pack, attention of H/P heads over all N keys, the K-blocked to_out) timed here with both all-to-alls
    replaced by the identity (UlyssesAttnProcessor(simulate_world=P)), plus an xGMI term from DESIGN section 6's bytes per
    layer: each rank sends 3 B (N/P) (D/P) 2 bytes to every peer and gets B (N/P) (D/P) 2 back, every peer pair on a link
    of its own at ~153 GB/s (MI355X_MICROARCH.md).  Latency, RCCL launch overhead and the overlap mode are not modelled."""
    from ltxmi import distributed as sp
    steps = max(3, min(steps, 10))
    out = {"label": "compute-only projection from one GPU, NOT a measurement (exchanges replaced by the identity; xGMI "
                    "term = bytes / 153 GB/s per link, no latency, no overlap)", "configs": {}}
    for name, grid, cond in (("config2_4992", GRID, False), ("config3_13376", GRID_CONFIG3, True)):
        runner.set_grid(grid, image_conditioned=cond)
        full = (runner.latents, runner.freqs, runner.t_dev, runner.grid, runner.n_tok)
        for _ in range(2):
            runner.step()
        e1, _ = timed_steps(runner.step, steps, None)
        t1 = e1 / steps * 1e3
        res = {"tokens": runner.n_tok, "t1_ms": round(t1, 2), "ranks": {}}
        for P in (2, 4, 8):
            nl = runner.n_tok // P
            if runner.n_tok % P or H % P or (cond and nl % (grid[1] * grid[2])):
                res["ranks"][str(P)] = {"skipped": "tokens / heads / whole frames not divisible"}
                continue
            for blk in runner.m.transformer_blocks:
                blk.attn1.set_processor(sp.UlyssesAttnProcessor(simulate_world=P))
            runner.latents = full[0][:, :nl].contiguous()
            runner.freqs = tuple(t[:, :nl].contiguous() for t in full[1])
            runner.t_dev = full[2][:, :nl].contiguous() if cond else full[2]
            runner.grid = (grid[0] // P, grid[1], grid[2]) if cond else grid
            try:
                for _ in range(2):
                    runner.step()
                eP, _ = timed_steps(runner.step, steps, None)
            finally:
                runner.latents, runner.freqs, runner.t_dev, runner.grid, runner.n_tok = full
                sp.disable_sequence_parallel(runner.m)
            tP = eP / steps * 1e3
            per_peer = (3 + 1) * NUM_CONDS * nl * (D // P) * 2                 # q,k,v out + o back, bytes to ONE peer per layer
            xgmi_ms = L * per_peer / 153e9 * 1e3
            res["ranks"][str(P)] = {"t_rank_ms": round(tP, 2), "compute_speedup": round(t1 / tP, 3),
                                    "compute_efficiency": round(t1 / tP / P, 3), "xgmi_ms_per_step": round(xgmi_ms, 3),
                                    "projected_speedup_with_exposed_exchange": round(t1 / (tP + xgmi_ms), 3)}
        out["configs"][name] = res
    runner.set_grid(GRID)
    return out


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]


def attention_band_check(q, k, v, out, what, rows=64):
    """A band of query rows against plain fp32 torch on the GPU (outside any timed region): the legs below must
    not report the time of a kernel that produced garbage."""
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite attention output"
    n = q.shape[1]
    for r0 in (0, max(0, n - rows)):
        qs = q[:, r0:r0 + rows].float().permute(0, 2, 1, 3)                  # [B,H,r,dh]
        sc = torch.matmul(qs, k.float().permute(0, 2, 3, 1)) * (q.shape[-1] ** -0.5)
        ref = torch.matmul(torch.softmax(sc, dim=-1), v.float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
        err = float((out[:, r0:r0 + rows].float() - ref).norm() / ref.norm())
        assert err < 2e-2, f"{what}: rows {r0}..{r0 + rows} differ from fp32 attention by {err:.3e}"


def time_attention_forms(device, iters=20):
    """The self-attention launch of the workload (B 3, H 32, N 4992, head_dim 64) in the kernel's two forms, and on two kinds of
    logits: N(0,1) q / k (what random-init weights produce: a few nats) and "trained-like" ones (every query aligned with its
    own token's key at ~+40 nats, everything else noise of ~6 nats: a nearly one-hot softmax).  steady = the normal run
    (reference-0 softmax, range-checked); exact = every item forced through the textbook online softmax (``force_exact``) --
    what the launch would cost if NOTHING fitted the +-100-bit range; redo_items = items the normal run had to redo."""
    from ltxmi import ops
    B, N = NUM_CONDS, N_TOK
    g = torch.Generator(device=device).manual_seed(7)

    def rnd(scale):
        return (torch.randn(B, N, H, DH, generator=g, device=device, dtype=torch.float32) * scale).to(torch.bfloat16)

    def ms(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    out = {"shape": [B, N, H, DH], "items": B * H * ((N + 255) // 256)}
    flop = 4.0 * B * H * N * N * DH
    q, k, v = rnd(1.0), rnd(1.0), rnd(1.0)
    q2 = rnd(2.0)
    qf = q2.float()
    k2 = (rnd(2.0).float() + qf * (40.0 * 8.0 / (qf * qf).sum(-1, keepdim=True))).to(torch.bfloat16)
    for name, (qq, kk) in (("random_logits", (q, k)), ("trained_like_logits", (q2, k2))):
        cnt = torch.zeros(1, dtype=torch.int32, device=device)
        o = ops.attention(qq, kk, v, redo_counter=cnt)
        oe = ops.attention(qq, kk, v, force_exact=True)
        assert torch.isfinite(o.float()).all() and torch.isfinite(oe.float()).all()
        d = float((o.float() - oe.float()).norm() / oe.float().norm())
        assert d < 6e-3, (name, d)                               # two bf16 renderings of the same softmax
        t_s, t_e = ms(lambda: ops.attention(qq, kk, v)), ms(lambda: ops.attention(qq, kk, v, force_exact=True))
        out[name] = {"steady_form_ms": round(t_s, 4), "exact_form_ms": round(t_e, 4), "redo_items": int(cnt.item()),
                     "steady_frac_of_mfma_peak": round(flop / (t_s * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "exact_frac_of_mfma_peak": round(flop / (t_e * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "steady_vs_exact_rel_l2": round(d, 5)}
    return out


def time_attention(device, n_tok, iters, B=1, heads=H, dh=DH, lk=None):
    from ltxmi import ops
    g = torch.Generator(device=device).manual_seed(3)
    qkv = torch.randn(B, n_tok, 3, heads, dh, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if lk is not None:                                       # cross-attention: lk keys
        k, v = k[:, :lk].contiguous(), v[:, :lk].contiguous()
    out = torch.empty(B, n_tok, heads, dh, device=device, dtype=torch.bfloat16)
    ops.attention(q, k, v, out=out)
    attention_band_check(q, k, v, out, f"attention N={n_tok} Lk={k.shape[1]} heads={heads} dh={dh}")
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention(q, k, v, out=out)
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ts]
    ms = pct(ts, 0.5)
    flops = 4.0 * B * n_tok * k.shape[1] * heads * dh       # QK^T + PV
    tf = flops / ms / 1e9
    res = {"tokens": n_tok, "keys": k.shape[1], "batch": B, "heads": heads, "head_dim": dh, "ms": round(ms, 3),
           "ms_p10": round(pct(ts, 0.1), 3), "ms_p90": round(pct(ts, 0.9), 3), "iters": iters,
           "tflops": round(tf, 1), "qk_frac_of_mfma_peak": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
           "algorithmic_bytes": 2 * B * heads * dh * (2 * n_tok + 2 * k.shape[1]),
           "note": "QK^T and PV run at the same rate: fraction = (4 Lq Lk H dh / t) / peak = (2 Lq Lk H dh / (t/2)) / peak"}
    D = heads * dh
    if lk is None and ops.attention_fuses_qnorm(B, heads, n_tok, n_tok, dh):
        # the same launch as Transformer3DModel.forward makes it at this shape: q is the raw projection output and the kernel
        # applies q_norm (row factor), its weight and RoPE while loading it (then softmax_scale * log2(e) goes into q before
        # its rounding and the loop has no multiply per score).  Checked against the two-pass form (q's pass, then the launch
        # timed above) on a band of rows.
        wq = (1.0 + 0.1 * torch.randn(D, generator=g, device=device)).to(torch.bfloat16)
        ang = torch.rand(n_tok, D // 2, generator=g, device=device) * 6.28
        cos, sin = (f(ang).repeat_interleave(2, dim=-1).to(torch.bfloat16) for f in (torch.cos, torch.sin))
        del ang
        q2 = q.reshape(B * n_tok, D)
        rstd = torch.rsqrt(q2.float().pow(2).mean(-1) + 1e-6).contiguous()
        out_f = torch.empty_like(out)
        ops.attention(q, k, v, out=out_f, q_norm=(rstd, wq, 1e-6), rope=(cos, sin, n_tok))
        band = slice(n_tok - 256, n_tok)
        qb = q[:, band].clone()
        ops.rmsnorm_rope_(qb.view(-1, D), wq, 1e-6, cos[band], sin[band], 256)
        two_pass = ops.attention(qb, k, v)
        err = float((out_f[:, band].float() - two_pass.float()).norm() / two_pass.float().norm())
        assert err < 6e-3, f"attention N={n_tok}: q on load differs from the two-pass form by {err:.3e}"
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.attention(q, k, v, out=out_f, q_norm=(rstd, wq, 1e-6), rope=(cos, sin, n_tok))
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        ms_f = pct([a.elapsed_time(b) for a, b in ts], 0.5)
        res["as_the_model_launches_it"] = {
            "ms": round(ms_f, 3), "tflops": round(flops / ms_f / 1e9, 1),
            "qk_frac_of_mfma_peak": round(flops / ms_f / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
            "note": "q_norm + weight + RoPE applied to q on load (attention.py:1040-1055 of the reference fused into the launch, "
                    "as ltxmi.Transformer3DModel does at this shape); same FLOP count, the q pass it replaces not credited"}
    return res


def make_vae(device, grid=GRID, z_tile=0):
    """The bench's decoder (0.9.5+-style timestep-conditioned, create_video_autoencoder_demo_config(128),
    causal_video_autoencoder.py:1302-1338, random-init bf16), a latent z [1,128,*grid] and timestep 0.05."""
    import ltxmi
    blocks = [("res_x", {"num_layers": 2, "inject_noise": False}), ("compress_all", {"residual": True, "multiplier": 2}),
              ("res_x", {"num_layers": 2, "inject_noise": False}), ("compress_all", {"residual": True, "multiplier": 2}),
              ("res_x", {"num_layers": 2, "inject_noise": False}), ("compress_all", {"residual": True, "multiplier": 2}),
              ("res_x", {"num_layers": 2, "inject_noise": False})]
    cfg = {"_class_name": "CausalVideoAutoencoder", "dims": 3, "decoder_blocks": blocks, "latent_channels": C_LAT,
           "norm_layer": "pixel_norm", "patch_size": 4, "latent_log_var": "uniform", "use_quant_conv": False,
           "causal_decoder": False, "timestep_conditioning": True, "spatial_padding_mode": "replicate"}
    # (decoder_base_channels defaults to 128: 1024 -> 512 -> 256 -> 128 feature channels, conv_out 128 -> 48)
    torch.manual_seed(5)
    with torch.device(device):
        vae = ltxmi.CausalVideoAutoencoder.from_config(dict(cfg))
    vae = vae.to(dtype=torch.bfloat16).eval()
    vae.decoder.timestep_scale_multiplier.data = vae.decoder.timestep_scale_multiplier.data.float()
    z = torch.randn(1, C_LAT, *grid, device=device).to(torch.bfloat16)
    ts = torch.tensor([0.05], device=device)
    if z_tile:
        vae.enable_z_tiling(z_tile)
    return vae, z, ts


HBM_PEAK_GBS = 8000.0                # HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)


def vae_stage_positions(grid):
    """Positions of the four decoder stages for a latent grid: (F,H,W) -> (2F-1, 2H, 2W) per depth-to-space upsample."""
    f, h, w = grid
    return [f * h * w, (2 * f - 1) * 4 * h * w, (4 * f - 3) * 16 * h * w, (8 * f - 7) * 64 * h * w]


def vae_decode_flop(grid):
    """sum over the convolutions of 2 * 27 * Cin * Cout * positions for the bench's decoder (make_vae):
    128->1024, 4x 1024->1024, 1024->4096 | 4x 512->512, 512->2048 | 4x 256->256, 256->1024 | 4x 128->128, 128->48."""
    s0, s1, s2, s3 = vae_stage_positions(grid)
    per_pos = [128 * 1024 + 4 * 1024 * 1024 + 1024 * 4096, 4 * 512 * 512 + 512 * 2048, 4 * 256 * 256 + 256 * 1024,
               4 * 128 * 128 + 128 * 48]
    return 54.0 * sum(p * c for p, c in zip((s0, s1, s2, s3), per_pos))


def _vae_traffic(tiled=False):
    tpath = os.path.join(ROOT, "profiles", "traffic_vae_config5.json" if tiled else "traffic_vae.json")
    try:
        return json.load(open(tpath))
    except Exception:  # noqa: BLE001
        return {}


def time_vae(device, iters, grid=GRID, z_tile=0, cross_check=True):
    """CausalVideoAutoencoder.decode of z [1,128,*grid] with the decoder of make_vae.
    z_tile > 0: the reference's z-tiling (vae.py:365-402, tiles of z_tile + 1 latent frames, blends).
    The untiled result is cross-checked against the same decode with every convolution as an implicit GEMM
    (a second, independently tested implementation) before anything is timed."""
    import ltxmi
    from ltxmi import ops
    vae, z, ts = make_vae(device, grid, z_tile)
    check = {}
    with torch.no_grad():
        img = ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
        assert torch.isfinite(img.float()).all(), "non-finite VAE decode output"
        if not cross_check:
            check = {"cross_check": "skipped (profiling run: the implicit-GEMM rendering would show up in the kernel statistics)"}
        elif not z_tile:
            old = ops.CONV_ALGO
            try:
                ops.CONV_ALGO = 1
                ref = ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
            finally:
                ops.CONV_ALGO = old
            err = float((img.float() - ref.float()).norm() / ref.float().norm())
            # two independent bf16 renderings, each ~8e-3 from fp32 truth: an indexing error would be 1e-1 .. 1
            # each path is within (bf16-eager error + 2e-3) ~ 1e-2 of the fp32 oracle at this very size
            # (tests/test_gpu_model.py::test_vae_decode_config2_full_size_vs_fp32_oracle): 2e-2 bounds their distance
            assert err < 2e-2, f"VAE decode: direct-convolution and implicit-GEMM paths differ by {err:.3e}"
            check = {"rel_l2_vs_implicit_gemm_decode": round(err, 5)}
            del ref
        else:
            # the tiled decode's frames before the first blend are tile 0's untiled decode, bit for bit; and the whole
            # clip agrees with its all-implicit-GEMM rendering (the checks of test_vae_decode_config5_full_size_tiled)
            tl = z_tile
            vae.disable_z_tiling()
            t0img = ltxmi.vae_decode(z[:, :, :tl + 1], vae, True, vae_per_channel_normalize=True, timestep=ts)
            vae.enable_z_tiling(z_tile)
            keep = 8 * tl - 2 * tl + 1
            assert torch.equal(img[:, :, :keep], t0img.to(img.dtype)[:, :, :keep]), "tiled decode: tile 0 frames differ"
            del t0img
            old = ops.CONV_ALGO
            try:
                ops.CONV_ALGO = 1
                ref = ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
            finally:
                ops.CONV_ALGO = old
            err = float((img.float() - ref.float()).norm() / ref.float().norm())
            assert err < 2e-2, f"tiled VAE decode: direct-convolution and implicit-GEMM paths differ by {err:.3e}"
            check = {"tile0_frames_bit_equal_untiled": True, "rel_l2_vs_implicit_gemm_decode": round(err, 5)}
            del ref
        torch.cuda.synchronize()
        # the dominant convolution launch (128 -> 128 at the full-resolution stage: 4 per decode / per tile)
        pos3 = vae_stage_positions((z_tile + 1,) + tuple(grid[1:]) if z_tile else grid)[3]
        # (conv1 of the two ResnetBlock3Ds there: the plain convolution + bias with norm2 -> SiLU in its epilogue; conv2's
        # launches carry the skip add and a second, activated output on top and are not averaged in)
        key_conv = ("conv3d", pos3, 128, 128, 0, "post_norm")
        ops.watch_launches([key_conv])
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            img = ltxmi.vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        conv_ms = ops.launch_times_ms().get(key_conv, [])
        ops.watch_launches(None)
    dt = pct(times, 0.5)
    frames = img.shape[2]
    out = {"latent": [1, C_LAT] + list(grid), "frames": frames, "shape": list(img.shape), "z_tile": z_tile,
           "ms_per_decode": round(dt * 1e3, 2), "ms_p10": round(pct(times, 0.1) * 1e3, 2),
           "ms_p90": round(pct(times, 0.9) * 1e3, 2), "iters": iters, "frames_per_s": round(frames / dt, 2)}
    if not z_tile:
        tf = vae_decode_flop(grid) / 1e12
        out.update({"algorithmic_tflop": round(tf, 2), "tflops": round(tf / dt, 1),
                    "frac_of_mfma_peak": round(tf / dt / MFMA_BF16_PEAK_TFLOPS, 4)})
    if conv_ms:
        ms = sum(conv_ms) / len(conv_ms)
        flop = 54.0 * 128 * 128 * pos3
        byts = (128 + 128) * pos3 * 2 + 27 * 128 * 128 * 2
        out["roofline_conv"] = {
            "kernel": f"conv3d_direct_v3_kernel<3> (CausalConv3d 128 -> 128 at {pos3} positions, the full-resolution stage; "
                      "norm2 -> AdaLN -> SiLU in its epilogue)",
            "bound": "mfma", "achieved": round(flop / ms / 1e9, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flop / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4), "launch_ms": round(ms, 4), "launches_timed": len(conv_ms),
            "algorithmic_flop_per_launch": flop, "algorithmic_bytes_per_launch": byts,
            "traffic": (_vae_traffic().get("conv_direct_hbm_bytes_per_launch") if (not z_tile and tuple(grid) == GRID) else
                        _vae_traffic(True).get("conv_direct_hbm_bytes_per_launch") if (z_tile == 4 and tuple(grid) == (33, 23, 40))
                        else None),
            "traffic_source": "static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/vae_time.py (tools/vae_tiled_time.py "
                              "for the z-tiled decode) committed as " +
                              _vae_traffic(bool(z_tile)).get("source", "profiles/traffic_vae.json") + " (gfx950 corrections applied)",
            "hbm_side": {"bound": "hbm", "achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4),
                         "note": "BASELINE.json calls this kernel HBM-bound; at 1728 FLOP/B it is not: both roofs reported"}}
    out.update(check)
    return out


def time_vae_tile_parallel(device, iters, dist, grid=(33, 23, 40), z_tile=4):
    """BASELINE config 5 over the ranks: the z-tiles of the tiled decode spread over the GPUs (tile n on rank n mod P,
    broadcast, blends everywhere: ltxmi.distributed.tile_parallel_vae_decode).  MAX over ranks of the median."""
    from ltxmi import distributed as sp
    vae, z, ts = make_vae(device, grid, z_tile)
    times = []
    with torch.no_grad():
        img = sp.tile_parallel_vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
        assert torch.isfinite(img.float()).all(), "non-finite tile-parallel VAE decode output"
        for _ in range(iters):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            img = sp.tile_parallel_vae_decode(z, vae, True, vae_per_channel_normalize=True, timestep=ts)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    dt = max_over_ranks(pct(times, 0.5), dist, device)
    frames = img.shape[2]
    return {"latent": [1, C_LAT] + list(grid), "frames": frames, "shape": list(img.shape), "z_tile": z_tile,
            "tiles_over_ranks": dist.get_world_size(), "ms_per_decode": round(dt * 1e3, 2), "iters": iters,
            "frames_per_s": round(frames / dt, 2)}


def _host_threads():
    # the GPU box shows every host core but a 1-GPU job's CPU share is 16 (oversubscribing all 256
    # logical cores made the oracle 6x slower): use the scheduler affinity, capped at 16
    ncpu = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(ncpu)
    return ncpu


def cpu_baseline(layers=L):
    """The oracle's fp32 restatement of Transformer3DModel.forward (oracle/dit.py) at the bench shape, measured DIRECTLY:
    the full forward -- patchify projection, AdaLN-single, caption projection, all 28 blocks, output head -- for ONE
    cond at N = 4992 on the host cores (about 25 s).  A B_eff = 3 step is three such forwards (the batch rows are
    independent and the CPU is compute-bound), + guidance / Euler (negligible).  To keep 7.7 GB of random fp32 weights
    out of the run, the 28 blocks alias ONE block's tensors: same shapes, same FLOP, same memory traffic per block."""
    from oracle import dit, sched
    ncpu = _host_threads()
    cfg1 = dict(dit.default_2b_config(), num_layers=1)
    sd = dit.init_state_dict(cfg1, seed=0)
    cfg = dict(cfg1, num_layers=layers)
    for k in [k for k in sd if k.startswith("transformer_blocks.0.")]:
        for i in range(1, layers):
            sd[k.replace("transformer_blocks.0.", f"transformer_blocks.{i}.", 1)] = sd[k]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, N_TOK, C_LAT, generator=g)
    enc = torch.randn(1, T_TEXT, 4096, generator=g)
    mask = torch.zeros(1, T_TEXT)
    mask[:, :96] = 1
    ts = torch.full((1, 1), 0.7)
    fc = dit.precompute_freqs_cis(sched.fractional_coords(*GRID, 1, 25.0), cfg, torch.float32)
    with torch.no_grad():
        dit.transformer_block(sd, "transformer_blocks.0.", cfg1, torch.randn(1, N_TOK, D, generator=g), fc,
                              torch.randn(1, T_TEXT, D, generator=g), torch.zeros(1, 1, T_TEXT),
                              torch.randn(1, 1, 6 * D, generator=g) * 0.1)                     # warm (threads, allocator)
        t0 = time.perf_counter()
        out = dit.transformer3d_forward(sd, cfg, x, fc, enc, ts, encoder_attention_mask=mask, latent_shape=GRID)
        fwd = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return {"value": round(1.0 / (fwd * NUM_CONDS), 6), "unit": "denoise-steps/s", "cores": ncpu, "kind": "port",
            "sample": f"ONE full {layers}-layer Transformer3DModel.forward of the fp32 oracle at N=4992, 1 cond, measured "
                      f"directly: {fwd:.1f} s; a B_eff={NUM_CONDS} step = {NUM_CONDS} such forwards (rows independent)",
            "forward_s_one_cond": round(fwd, 2)}


def cpu_baseline_vae():
    """The oracle's fp32 VAE decode (oracle/vae.py) on the host cores: BASELINE config 1's decode (z [1,128,2,8,8] -> 9
    frames of 256 x 256) measured whole, and a 3-latent-frame slab of config 2 (z [1,128,3,16,24], ~5 TFLOP) measured
    whole; config 2's frames/s is extrapolated from the slab by algorithmic FLOP and labelled so."""
    from oracle import vae as ov
    ncpu = _host_threads()
    cfg = ov.demo_config(C_LAT)
    cfg["decoder_base_channels"] = 128
    cfg["build_encoder"] = False
    sd = ov.init_state_dict(cfg, seed=0)
    ts = torch.tensor([0.05])
    g = torch.Generator().manual_seed(0)
    res = {}
    with torch.no_grad():
        z1 = torch.randn(1, C_LAT, 2, 8, 8, generator=g)
        ov.vae_decode(sd, cfg, z1, ts)                                                          # warm
        t0 = time.perf_counter()
        img = ov.vae_decode(sd, cfg, z1, ts)
        t1 = time.perf_counter() - t0
        res["config1_decode"] = {"latent": [1, C_LAT, 2, 8, 8], "frames": img.shape[2], "s": round(t1, 2),
                                 "frames_per_s": round(img.shape[2] / t1, 3), "tflop": round(vae_decode_flop((2, 8, 8)) / 1e12, 3)}
        slab = (3, GRID[1], GRID[2])
        z2 = torch.randn(1, C_LAT, *slab, generator=g)
        t0 = time.perf_counter()
        img = ov.vae_decode(sd, cfg, z2, ts)
        t2 = time.perf_counter() - t0
    f_slab, f_full = vae_decode_flop(slab), vae_decode_flop(GRID)
    full_s = t2 * f_full / f_slab
    frames = 8 * (GRID[0] - 1) + 1
    res.update({"value": round(frames / full_s, 4), "unit": "VAE-decode frames/s (768x512x97)", "cores": ncpu, "kind": "port",
                "extrapolated": True,
                "sample": f"fp32 oracle decode of z [1,128,{slab[0]},{slab[1]},{slab[2]}] ({f_slab / 1e12:.2f} TFLOP) measured whole: "
                          f"{t2:.1f} s = {f_slab / t2 / 1e12:.3f} TFLOP/s; config 2 ({f_full / 1e12:.1f} TFLOP, {frames} frames) scaled "
                          f"by FLOP -> {full_s:.0f} s"})
    return res


def timed_steps(step_fn, steps, dist):
    """EXACTLY `steps` calls of step_fn bracketed by barrier + synchronize on both sides; returns (elapsed seconds by
    the host clock, per-step milliseconds from HIP events on the launch stream)."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step_fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def _stage_collectives_through_host(dist):
    """Rehearsal only (LTXMI_BENCH_REHEARSAL=gloo): gloo moves CPU tensors, so the collectives ltxmi.distributed and this
    file issue on device tensors are staged through the host."""
    a2a, gather, reduce = dist.all_to_all_single, dist.all_gather, dist.all_reduce

    def all_to_all_single(recv, send, group=None, **kw):
        if not send.is_cuda:
            return a2a(recv, send, group=group, **kw)
        r = torch.empty(recv.shape, dtype=recv.dtype)
        a2a(r, send.cpu(), group=group, **kw)
        recv.copy_(r)

    def all_gather(parts, x, group=None, **kw):
        if not x.is_cuda:
            return gather(parts, x, group=group, **kw)
        host = [torch.empty(t.shape, dtype=t.dtype) for t in parts]
        gather(host, x.cpu(), group=group, **kw)
        for t, h in zip(parts, host):
            t.copy_(h)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, **kw):
        if not t.is_cuda:
            return reduce(t, op=op, group=group, **kw)
        h = t.cpu()
        reduce(h, op=op, group=group, **kw)
        t.copy_(h)

    bcast = dist.broadcast

    def broadcast(t, src=0, group=None, **kw):
        if not t.is_cuda:
            return bcast(t, src=src, group=group, **kw)
        h = t.cpu()
        bcast(h, src=src, group=group, **kw)
        t.copy_(h)

    gather_flat = dist.all_gather_into_tensor

    def all_gather_into_tensor(out, x, group=None, **kw):
        if not x.is_cuda:
            return gather_flat(out, x, group=group, **kw)
        h = torch.empty(out.shape, dtype=out.dtype)
        gather_flat(h, x.cpu(), group=group, **kw)
        out.copy_(h)

    dist.all_to_all_single, dist.all_gather, dist.all_reduce, dist.broadcast = all_to_all_single, all_gather, all_reduce, broadcast
    dist.all_gather_into_tensor = all_gather_into_tensor


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher in the environment: run the driver's own N > 1 command
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)
    as a CHILD process and return its exit code.  Nothing in this process has touched the GPU (torch.cuda.device_count()
    does not initialise it on this image), so no initialised process is ever replaced or forked.  A box with fewer GPUs
    than ranks is refused (exit 2) unless LTXMI_BENCH_REHEARSAL=gloo: never an `n_gpus: 1` line for `--gpus N`."""
    import socket
    import subprocess
    if not os.environ.get("LTXMI_BENCH_REHEARSAL") and "--selftest-launch" not in argv:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but this box shows {have} GPU(s); not running (set LTXMI_BENCH_REHEARSAL=gloo to "
                  f"rehearse the control flow with the ranks sharing the visible GPUs)", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd).returncode


def selftest_launch(rank, world):
    """What `--selftest-launch` runs in every rank: the rendezvous the launcher set up works, and rank 0 prints one line."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world + 1) / 2 and dist.get_world_size() == world
    if rank == 0:
        print(json.dumps({"selftest_launch": bool(ok), "n_gpus": world}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def agree_failed(err, dist, device):
    """a leg that failed on ONE rank (e.g. an allocation failure) must be abandoned by all of them together --
    MAX all-reduce of a failure flag after the leg, so that nobody goes on to a collective the failing rank never joins."""
    if dist is None:
        return err is not None
    t = torch.tensor([1 if err is not None else 0], device=device, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def max_over_ranks(x, dist, device):
    if dist is None:
        return x
    t = torch.tensor([x], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true", help="skip the B_eff-1 / attention-shape / VAE / CPU legs")
    ap.add_argument("--parallelism", choices=["both", "replicas", "ulysses"], default="both",
                    help="N > 1 only.  replicas: every rank denoises its own video, no data-path collective (weak scaling, "
                         "this is `value`); ulysses: ONE video, tokens sharded over the ranks, all-to-all inside "
                         "self-attention (ltxmi.distributed, strong scaling); both (default): `value` from replicas and "
                         "the Ulysses run reported in the same line under \"ulysses\"")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="N > 1: seconds the legs that run collectives (Ulysses at two sizes, the tile-parallel VAE decode) may "
                         "take together before a watchdog prints the line with the replicas measurement and an error entry for "
                         "them, and ends the ranks (they could only be run at world size 1 on hardware so far: a hang there "
                         "must not take the headline value with it)")
    ap.add_argument("--rehearse-both", action="store_true",
                    help="run the N > 1 control flow (replicas, then Ulysses, both in the one JSON line) at world size 1")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="launcher check (CPU suite): every rank joins a gloo group, rank 0 prints {\"selftest_launch\": true, "
                         "\"n_gpus\": world}; no GPU is touched")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE anything here touches the
        # GPU (this process never initialises HIP; it only waits for its child and passes its exit code on)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line's n_gpus would not be what was asked for")
    if args.selftest_launch:
        return selftest_launch(rank, world)
    # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the build boxes have one): LTXMI_BENCH_REHEARSAL=gloo
    # lets the ranks share the visible GPUs and carries the collectives over gloo, staged through the host.  The numbers
    # of such a run mean nothing (the ranks time-share a GPU) and the line says so; what it checks is the code path.
    rehearsal = os.environ.get("LTXMI_BENCH_REHEARSAL", "")
    if rehearsal == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    ulysses_only = args.parallelism == "ulysses"
    if world > 1 or ulysses_only or args.rehearse_both:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a five-line version banner on STDOUT when the first communicator is created; rank 0's stdout
        # must carry the one JSON line only, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal == "gloo":
                dist.init_process_group("gloo")
                _stage_collectives_through_host(dist)
            else:
                dist.init_process_group("nccl", device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from ltxmi import ops
    ops.set_step_invariant_caching(False)     # headline: every timed step does all the work of the reference's step
    runner = StepRunner(device)

    def measure(ulysses):
        sp = world if ulysses else 1                          # token (and, inside attention, head) shards
        M = NUM_CONDS * runner.n_tok // sp
        key_ff1 = ("gemm", M, FF, D, ops.EPI_GELU_TANH)
        key_attn = ("attention", NUM_CONDS, H // sp, runner.n_tok, runner.n_tok, DH)
        for _ in range(args.warmup):
            runner.step()
        redo = torch.zeros(1, dtype=torch.int32, device=device)
        ops.count_attention_redos(redo)          # items of the self- / cross-attention launches redone in the exact form
        ops.watch_launches([key_ff1, key_attn])
        elapsed, per_step = timed_steps(runner.step, args.steps, dist)
        times = ops.launch_times_ms()
        ops.watch_launches(None)
        ops.count_attention_redos(None)
        elapsed = max_over_ranks(elapsed, dist, device)
        assert torch.isfinite(runner.latents).all(), "non-finite latents after the timed steps"
        ff1, at = times.get(key_ff1, []), times.get(key_attn, [])
        ff1_ms, at_ms = sum(ff1) / max(len(ff1), 1), sum(at) / max(len(at), 1)
        return {"elapsed": elapsed, "per_step": per_step, "sp": sp, "M": M, "ff1_ms": ff1_ms, "n_ff1": len(ff1),
                "at_ms": at_ms, "n_at": len(at), "redo_items": int(redo.item())}

    line = None

    def build_line():
        """rank 0: the line's headline part from the measurement r (everything the contract asks for)."""
        ms_per_step = r["elapsed"] / args.steps * 1e3
        value = (1 if ulysses_only else world) * args.steps / r["elapsed"]
        sp, M = r["sp"], r["M"]
        ff1_flop = 2.0 * M * FF * D
        ff1_tf = ff1_flop / (r["ff1_ms"] * 1e-3) / 1e12 if r["ff1_ms"] > 0 else 0.0
        at_flop = 4.0 * NUM_CONDS * N_TOK * N_TOK * (D // sp)
        at_tf = at_flop / (r["at_ms"] * 1e-3) / 1e12 if r["at_ms"] > 0 else 0.0
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        tnote = ("static: HBM-side bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                 "command committed as " + traffic.get("source", "profiles/traffic.json") +
                 " (gfx950 corrections applied); not re-measured in this run")
        line = {
            "metric": "denoise-steps/sec + VAE-decode frames/sec, LTX-Video 768x512x97f at 1/8 GPU",
            "value": round(value, 4), "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong" if ulysses_only else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "LTX-Video 2B t2v 768x512x97, one denoise step = Transformer3DModel.forward "
                                   "(28 layers, D 2048, 32x64 heads, N 4992 tokens, T 256) at B_eff 3 "
                                   "(CFG + STG rows) + fused guidance/Euler; random-init weights",
                       "tokens": N_TOK, "b_eff": NUM_CONDS, "layers": L,
                       "parallelism": (f"ulysses sp{world}" if ulysses_only else f"replicas x{world}"),
                       "algorithmic_tflop_per_step": 66.4},
            "step_tflops": round(66.4 / (ms_per_step * 1e-3), 1),
            "step_ms": {"median": round(pct(r["per_step"], 0.5), 3), "p10": round(pct(r["per_step"], 0.1), 3),
                        "p90": round(pct(r["per_step"], 0.9), 3), "n": len(r["per_step"]),
                        "note": "HIP events around each timed step on the launch stream"},
            "roofline": {"kernel": f"attention (ltxmi::pipe::attn_pipe_kernel, self-attention B={NUM_CONDS} H={H // sp} "
                                   f"N={N_TOK} dh={DH}; 28 launches per step, the largest single share of a step)",
                         "bound": "mfma", "achieved": round(at_tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(at_tf / MFMA_BF16_PEAK_TFLOPS, 4),
                         "traffic": traffic.get("attention_hbm_bytes_per_launch"), "traffic_source": tnote,
                         # north star: "achieved HBM GB/s on attention" = PMC bytes per launch over this run's launch time
                         "hbm_gb_per_s": (round(traffic["attention_hbm_bytes_per_launch"] / (r["at_ms"] * 1e-3) / 1e9, 1)
                                          if traffic.get("attention_hbm_bytes_per_launch") and r["at_ms"] > 0 else None),
                         "launch_ms": round(r["at_ms"], 4), "launches_timed": r["n_at"],
                         # the kernel's normal run is the reference-0 ("steady") softmax form, valid while scaled scores stay
                         # within ~+-100 bits; items that leave the range are detected and redone in the exact form
                         "redo_items": r["redo_items"],
                         "items_per_launch": NUM_CONDS * (H // sp) * ((N_TOK + 255) // 256),
                         "redo_note": "items (batch, head, 256-query tile) of ALL attention launches of the timed steps that "
                                      "were redone in the exact online-softmax form (device counter); random-init weights give "
                                      "logits of a few nats -- see attention.workload_exact_form for what the exact form costs",
                         "algorithmic_flop_per_launch": at_flop,
                         "algorithmic_bytes_per_launch": 8 * NUM_CONDS * N_TOK * (D // sp)},
            "roofline_gemm": {"kernel": f"gemm_bf16_nt_persistent_kernel<256,256,2,4,GELU_TANH> (ff.net.0, M={M} N=8192 K=2048)",
                              "bound": "mfma", "achieved": round(ff1_tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ff1_tf / MFMA_BF16_PEAK_TFLOPS, 4),
                              "traffic": traffic.get("ff1_gemm_hbm_bytes_per_launch"), "traffic_source": tnote,
                              "launch_ms": round(r["ff1_ms"], 4), "launches_timed": r["n_ff1"],
                              "algorithmic_flop_per_launch": ff1_flop},
        }
        return line

    def put_ulysses(line, uly):
        if "error" in uly:
            line["ulysses"] = {"error": uly["error"], "parallelism": f"ulysses sp{world}"}
            return
        ums = uly["elapsed"] / args.steps * 1e3
        line["ulysses"] = {"value": round(args.steps / uly["elapsed"], 4), "unit": "denoise-steps/s (ONE video)",
                           "scaling": "strong", "ms_per_step": round(ums, 2), "parallelism": f"ulysses sp{world}",
                           "step_ms": {"median": round(pct(uly["per_step"], 0.5), 3),
                                       "p10": round(pct(uly["per_step"], 0.1), 3),
                                       "p90": round(pct(uly["per_step"], 0.9), 3)},
                           "attention_launch_ms": round(uly["at_ms"], 4),
                           "note": "same step, tokens sharded N/P per rank, packed q,k,v all-to-all + o all-to-all per "
                                   "layer over RCCL, final all-gather (ltxmi/distributed.py)"}

    def put_ulysses_config3(line, uly3):
        n3 = GRID_CONFIG3[0] * GRID_CONFIG3[1] * GRID_CONFIG3[2]
        if "error" in uly3:
            line["ulysses_config3"] = {"error": uly3["error"]}
            return
        line["ulysses_config3"] = {
            "workload": "LTX-Video 2B i2v 1216x704x121 (13376 tokens, first latent frame conditioned: per-token timesteps), "
                        "B_eff 3, one video over the ranks", "tokens": n3, "parallelism": f"ulysses sp{world}",
            "value": round(args.steps / uly3["elapsed"], 4), "unit": "denoise-steps/s (ONE video)", "scaling": "strong",
            "ms_per_step": round(uly3["elapsed"] / args.steps * 1e3, 2),
            "step_ms": {"median": round(pct(uly3["per_step"], 0.5), 3), "p10": round(pct(uly3["per_step"], 0.1), 3),
                        "p90": round(pct(uly3["per_step"], 0.9), 3)},
            "attention_launch_ms": round(uly3["at_ms"], 4)}

    if ulysses_only:
        runner.enable_ulysses()
        r = measure(True)
        if rank == 0:
            line = build_line()
    else:
        r = measure(False)
        if rank == 0:
            line = build_line()
        if (world > 1 or args.rehearse_both) and args.parallelism == "both":
            # The replicas measurement above is the line's `value`.  The legs below run collectives on paths that could only be
            # run at world size 1 on hardware so far.  Two nets under them: an exception between collectives is agreed among
            # the ranks and reported (agree_failed); a HANG inside a collective ends in the watchdog, which prints the line as it
            # stands (rank 0 fills it leg by leg) with an error entry and ends this rank -- every rank runs its own, with the
            # same deadline.
            done = {"legs": []}

            def expired():
                msg = (f"watchdog: the collective legs did not finish within {args.collective_timeout:g} s "
                       f"(finished: {done['legs'] or 'none'}); the replicas measurement is unaffected")
                sys.stderr.write(f"[bench rank {rank}] {msg}\n")
                sys.stderr.flush()
                if rank == 0:
                    for key in ("ulysses", "ulysses_config3", "vae_decode_config5_tile_parallel"):
                        if key not in line and not (key.startswith("vae") and args.no_extras):
                            line[key] = {"error": msg}
                    print(json.dumps(line), flush=True)
                os._exit(0)

            watchdog = threading.Timer(args.collective_timeout, expired)
            watchdog.daemon = True
            watchdog.start()
            err = None
            try:
                runner.enable_ulysses()
                uly = measure(True)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"[:400]
            if agree_failed(err, dist, device):
                uly = {"error": err or "another rank failed in the Ulysses leg"}
            if rank == 0:
                put_ulysses(line, uly)
            done["legs"].append("ulysses")
            # BASELINE config 3, the workload the Ulysses mode is named for: 2B i2v 1216x704x121 = 16 x 22 x 38 = 13376
            # tokens, first latent frame conditioned (per-token timesteps), one video over the ranks
            if "error" not in uly:
                err = None
                uly3 = None
                try:
                    runner.set_grid(GRID_CONFIG3, image_conditioned=True)
                    uly3 = measure(True)
                except Exception as e:  # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"[:400]
                if agree_failed(err, dist, device):
                    uly3 = {"error": err or "another rank failed in the Ulysses config-3 leg"}
                runner.set_grid(GRID)
                if rank == 0:
                    put_ulysses_config3(line, uly3)
                done["legs"].append("ulysses_config3")
            runner.sp = None
            if not args.no_extras:
                # the VAE half of the metric over the ranks (every rank takes part: collective)
                try:
                    vtp = time_vae_tile_parallel(device, 3, dist)
                except Exception as e:  # noqa: BLE001
                    vtp = {"error": f"{type(e).__name__}: {e}"[:400]}
                if rank == 0:
                    line["vae_decode_config5_tile_parallel"] = vtp
                done["legs"].append("vae_decode_config5_tile_parallel")
            watchdog.cancel()
    extras = (not args.no_extras) and world == 1 and not ulysses_only
    if extras:
        # B_eff = 1 (report both): the same model, the text row only
        for _ in range(2):
            runner.step_one_cond()
        e1, per1 = timed_steps(runner.step_one_cond, args.steps, None)
        line["b_eff_1"] = {"denoise_steps_per_s": round(args.steps / e1, 4), "ms_per_step": round(e1 / args.steps * 1e3, 2),
                           "step_ms": {"median": round(pct(per1, 0.5), 3), "p10": round(pct(per1, 0.1), 3),
                                       "p90": round(pct(per1, 0.9), 3)},
                           "algorithmic_tflop_per_step": 22.1}
        # NOT the headline number: the same step with the STG "perturbed" row taken as a copy of the text row
        # for the 19 blocks before its first skipped block (bit-identical output, tests/test_gpu_model.py);
        # ltxmi.LTXVideoPipeline does this by default (stg_row_dedup)
        ops.set_step_invariant_caching(True)
        runner.step(stg_alias_blocks=19)
        ed, _ = timed_steps(lambda: runner.step(stg_alias_blocks=19), args.steps, None)
        dd = ed / args.steps
        line["pipeline_default"] = {"denoise_steps_per_s": round(1.0 / dd, 4), "ms_per_step": round(dd * 1e3, 2),
                                    "note": "as ltxmi.LTXVideoPipeline runs a step: STG row taken from the text row "
                                            "before its first skipped block + prompt K/V projected once per "
                                            "generation (both exact); not part of `value`"}
        ops.set_step_invariant_caching(False)
        line["config3_step_1gpu"] = time_config3_step(runner, ops, args.steps)
        line["ulysses_projection"] = ulysses_projection(runner, ops, args.steps)
        del runner
        torch.cuda.empty_cache()
        line["attention"] = {
            "workload_exact_form": time_attention_forms(device),
            "stress_98304": time_attention(device, 98304, 3),
            "config3_tokens_13376": time_attention(device, 13376, 20),
            "config4_wan_self_32760x12x128": time_attention(device, 32760, 20, heads=12, dh=128),
            "config4_wan_cross_512_keys": time_attention(device, 32760, 20, heads=12, dh=128, lk=512)}
        line["vae_decode"] = time_vae(device, 20)
        line["vae_decode_config5"] = time_vae(device, 5, grid=(33, 23, 40), z_tile=4)
        line["cpu_baseline"] = cpu_baseline()
        line["cpu_baseline_vae"] = cpu_baseline_vae()
    if rank == 0:
        if rehearsal:
            line["rehearsal"] = ("NOT a measurement: LTXMI_BENCH_REHEARSAL=" + rehearsal + ", the ranks share the visible "
                                 "GPUs and the collectives are staged through the host")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
