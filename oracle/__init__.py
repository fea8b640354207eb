"""CPU oracle for the LTX-Video denoise hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, any float dtype; fp32 = "truth", bf16 = the
reference's eager rounding points) restatement of the algorithm that the reference
project (soasme/LTX-Video-GPUPoor) runs on its hot path:

    Transformer3DModel.forward           ltx_video/models/transformers/transformer3d.py:328-507
    BasicTransformerBlock / Attention    ltx_video/models/transformers/attention.py:205-364, 986-1173
    pay_attention (sdpa eager branch)    wan/modules/attention.py:99-116, 162-199, 344-347
    CausalVideoAutoencoder.decode        ltx_video/models/autoencoders/{vae,causal_video_autoencoder,...}.py
    RectifiedFlowScheduler               ltx_video/schedulers/rf.py
    guidance math of the denoise loop    ltx_video/pipelines/pipeline_ltx_video.py:1183-1222
and, for the rows SURVEY.md 8f marks "next":
    Encoder / SpaceToDepthDownsample / encode / vae_encode     oracle/vae_encoder.py   (golden G11)
    prepare_conditioning, masked denoising_step, cond. noise   oracle/conditioning.py  (golden G12)
    LatentUpsampler, adain_filter_latent, _upsample_latents    oracle/upsampler.py     (golden G13)
    retrieve_timesteps, prepare_latents, guidance tables       oracle/pipeline_ctl.py  (golden G14; tables: golden G7)

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  The product package (``ltx-video-gpupoor_amd/ltxmi``) must
never import, call or fall back to anything in here; it fails loudly when the HIP
library is missing.

Pinning status
--------------
* Everything that is the reference's OWN code (the list above) is pinned: the
  golden vectors in ``tests/golden/*.safetensors`` were produced by importing the
  reference's modules from /root/reference in the build container
  (``oracle/gen/make_golden.py``) and ``tests/test_oracle_golden.py`` checks this
  restatement against them.
* The leaves the reference takes from the third-party ``diffusers`` package
  (>=0.31.0, not installed here, not vendored in the reference):
  ``AdaLayerNormSingle``, ``PixArtAlphaCombinedTimestepSizeEmbeddings``,
  ``TimestepEmbedding``, ``PixArtAlphaTextProjection``, ``RMSNorm``, ``GELU`` are
  restated in ``oracle/leaves.py`` from their published definitions.  The
  reference holds no test or fixture at that boundary, so those few functions
  are **parity unpinned** (the golden generator had to use the same restatement
  as a stand-in for the absent package).  The reference's own in-repo duplicate of
  the sinusoid (``ltx_video/models/transformers/embeddings.py:10-50``) does pin
  ``get_timestep_embedding``.
* Two more diffusers leaves were stood in for when generating G11-G14 (``oracle/gen``):
  ``DiagonalGaussianDistribution`` (mean / clamped logvar; its ``sample()`` was replaced by the mean
  because the reference draws unseeded noise there) and ``randn_tensor`` (= ``torch.randn``; the
  draws are stored in the fixtures and replayed).  Same status: **parity unpinned** at that boundary.
* The guidance math (CFG-star / STG / std-rescale, :1183-1222), the per-step guidance tables (:959-1013) and the
  loop plumbing of ``LTXVideoPipeline.__call__`` are pinned since round 2 by golden G7 (= SURVEY G7 + G11): the
  reference's own ``__call__`` run for config 1 (256x256x9, 2 steps, fp32, CPU) on an ``__init__``-less instance; its
  one GPU-only line, ``negative_prompt_attention_mask.to("cuda")`` (:1041), is satisfied by handing the mask in as a
  Tensor subclass whose ``.to("cuda")`` stays on the CPU (``oracle/gen/make_golden.py::g7``; nothing of the reference
  is edited).  ``guidance`` is written for one prompt (B = 1), which is all the reference's CFG-star line is
  well-formed for: its ``alpha * noise_pred_uncond`` multiplies ``[B, 1]`` by ``[B, N, C]`` (:1199), which for B > 1
  broadcasts alpha along the TOKEN axis (or fails) instead of per sample; the oracle's ``alpha.view(B, 1, 1)`` is the
  per-sample reading and coincides with the reference at B = 1 only.
"""
