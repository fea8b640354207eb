"""Checkpoint loading for the drop-in modules.

Formats the reference accepts (no mmgp, plain ``safetensors``; weights go straight to HBM):
  * single-file ``.safetensors`` whose header metadata carries ``config`` = JSON with
    ``transformer`` / ``vae`` / ``scheduler`` sections
      transformer3d.py:313-326, causal_video_autoencoder.py:103-115, rf.py:271-280
  * a diffusers directory (``transformer/config.json`` + ``diffusion_pytorch_model*.safetensors``,
    ``vae/...``): only the published Lightricks/LTX-Video configs are recognised and mapped to the
    native config, and parameter names are renamed
      transformer3d.py:278-312, causal_video_autoencoder.py:74-101,
      ltx_video/utils/diffusers_config_mapping.py:12-174
"""
import glob
import json
import os

import torch
from safetensors import safe_open

# ltx_video/utils/diffusers_config_mapping.py:140-145
TRANSFORMER_KEYS_RENAME = {"proj_in": "patchify_proj", "time_embed": "adaln_single",
                           "norm_q": "q_norm", "norm_k": "k_norm"}
# ltx_video/utils/diffusers_config_mapping.py:148-174 (order matters: applied sequentially)
VAE_KEYS_RENAME = {
    "decoder.up_blocks.3.conv_in": "decoder.up_blocks.7",
    "decoder.up_blocks.3.upsamplers.0": "decoder.up_blocks.8",
    "decoder.up_blocks.3": "decoder.up_blocks.9",
    "decoder.up_blocks.2.upsamplers.0": "decoder.up_blocks.5",
    "decoder.up_blocks.2.conv_in": "decoder.up_blocks.4",
    "decoder.up_blocks.2": "decoder.up_blocks.6",
    "decoder.up_blocks.1.upsamplers.0": "decoder.up_blocks.2",
    "decoder.up_blocks.1": "decoder.up_blocks.3",
    "decoder.up_blocks.0": "decoder.up_blocks.1",
    "decoder.mid_block": "decoder.up_blocks.0",
    "encoder.down_blocks.3": "encoder.down_blocks.8",
    "encoder.down_blocks.2.downsamplers.0": "encoder.down_blocks.7",
    "encoder.down_blocks.2": "encoder.down_blocks.6",
    "encoder.down_blocks.1.downsamplers.0": "encoder.down_blocks.4",
    "encoder.down_blocks.1.conv_out": "encoder.down_blocks.5",
    "encoder.down_blocks.1": "encoder.down_blocks.3",
    "encoder.down_blocks.0.conv_out": "encoder.down_blocks.2",
    "encoder.down_blocks.0.downsamplers.0": "encoder.down_blocks.1",
    "encoder.down_blocks.0": "encoder.down_blocks.0",
    "encoder.mid_block": "encoder.down_blocks.9",
    "conv_shortcut.conv": "conv_shortcut",
    "resnets": "res_blocks",
    "norm3": "norm3.norm",
    "latents_mean": "per_channel_statistics.mean-of-means",
    "latents_std": "per_channel_statistics.std-of-means",
}
# the one diffusers transformer config the reference recognises (diffusers_config_mapping.py:28-46)
DIFFUSERS_TRANSFORMER_CONFIG = {
    "_class_name": "LTXVideoTransformer3DModel", "_diffusers_version": "0.32.0.dev0",
    "activation_fn": "gelu-approximate", "attention_bias": True, "attention_head_dim": 64,
    "attention_out_bias": True, "caption_channels": 4096, "cross_attention_dim": 2048, "in_channels": 128,
    "norm_elementwise_affine": False, "norm_eps": 1e-06, "num_attention_heads": 32, "num_layers": 28,
    "out_channels": 128, "patch_size": 1, "patch_size_t": 1, "qk_norm": "rms_norm_across_heads",
}
# ... and what it maps to (OURS_TRANSFORMER_CONFIG, diffusers_config_mapping.py:74-105)
NATIVE_2B_TRANSFORMER_CONFIG = {
    "activation_fn": "gelu-approximate", "attention_bias": True, "attention_head_dim": 64,
    "attention_type": "default", "caption_channels": 4096, "cross_attention_dim": 2048,
    "double_self_attention": False, "dropout": 0.0, "in_channels": 128, "norm_elementwise_affine": False,
    "norm_eps": 1e-06, "norm_num_groups": 32, "num_attention_heads": 32, "num_embeds_ada_norm": 1000,
    "num_layers": 28, "num_vector_embeds": None, "only_cross_attention": False, "out_channels": 128,
    "upcast_attention": False, "use_linear_projection": False, "qk_norm": "rms_norm",
    "standardization_norm": "rms_norm", "positional_embedding_type": "rope",
    "positional_embedding_theta": 10000.0, "positional_embedding_max_pos": [20, 2048, 2048],
    "timestep_scale_multiplier": 1000,
}
# diffusers VAE config (diffusers_config_mapping.py:47-60) -> OURS_VAE_CONFIG (:106-130)
DIFFUSERS_VAE_CONFIG = {
    "_class_name": "AutoencoderKLLTXVideo", "_diffusers_version": "0.32.0.dev0",
    "block_out_channels": [128, 256, 512, 512], "decoder_causal": False, "encoder_causal": True,
    "in_channels": 3, "latent_channels": 128, "layers_per_block": [4, 3, 3, 3, 4], "out_channels": 3,
    "patch_size": 4, "patch_size_t": 1, "resnet_norm_eps": 1e-06,
}
NATIVE_VAE_CONFIG = {
    "_class_name": "CausalVideoAutoencoder", "dims": 3, "in_channels": 3, "out_channels": 3,
    "latent_channels": 128,
    "blocks": [["res_x", 4], ["compress_all", 1], ["res_x_y", 1], ["res_x", 3], ["compress_all", 1],
               ["res_x_y", 1], ["res_x", 3], ["compress_all", 1], ["res_x", 3], ["res_x", 4]],
    "scaling_factor": 1.0, "norm_layer": "pixel_norm", "patch_size": 4, "latent_log_var": "uniform",
    "use_quant_conv": False, "causal_decoder": False,
}


def _read_safetensors(path, device):
    tensors = {}
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata() or {}
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, meta


def _rename(sd, table):
    out = {}
    for k, v in sd.items():
        for a, b in table.items():
            k = k.replace(a, b)
        out[k] = v
    return out


def load_transformer(path, device="cuda", dtype=torch.bfloat16):
    """Build ``Transformer3DModel`` from a checkpoint (see the module docstring for the formats)."""
    from .transformer3d import Transformer3DModel
    path = str(path)
    if os.path.isdir(path):
        cfg = json.load(open(os.path.join(path, "transformer", "config.json")))
        if cfg != DIFFUSERS_TRANSFORMER_CONFIG:
            raise ValueError("Provided diffusers checkpoint config for transformer is not supported. "
                             "We only support diffusers configs found in Lightricks/LTX-Video.")
        config = dict(NATIVE_2B_TRANSFORMER_CONFIG)
        sd = {}
        for fpath in sorted(glob.glob(os.path.join(path, "transformer", "diffusion_pytorch_model*.safetensors"))):
            part, _ = _read_safetensors(fpath, device)
            sd.update(part)
        sd = _rename(sd, TRANSFORMER_KEYS_RENAME)
        strict = True
    elif path.endswith(".safetensors"):
        sd, meta = _read_safetensors(path, device)
        config = json.loads(meta["config"])["transformer"]
        strict = False        # single files also carry vae./text-encoder tensors (load_state_dict filters the prefix)
    else:
        raise ValueError(f"unrecognised checkpoint path: {path}")
    with torch.device("meta"):
        model = Transformer3DModel.from_config(config)
    if any(k.startswith("model.diffusion_model.") for k in sd):
        sd = {k.replace("model.diffusion_model.", ""): v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    sd = {k: v.to(dtype) for k, v in sd.items() if k in model.state_dict()} if not strict else \
        {k: v.to(dtype) for k, v in sd.items()}
    model.load_state_dict(sd, strict=True, assign=True)
    return model.eval()


def load_vae(path, device="cuda", dtype=torch.bfloat16, with_encoder=True):
    """Build ``CausalVideoAutoencoder`` from a checkpoint (``with_encoder=False``: decode side only)."""
    from .autoencoder import CausalVideoAutoencoder
    path = str(path)
    if os.path.isdir(path):
        cfg = json.load(open(os.path.join(path, "vae", "config.json")))
        if cfg != DIFFUSERS_VAE_CONFIG:
            raise ValueError("Provided diffusers checkpoint config for VAE is not supported. "
                             "We only support diffusers configs found in Lightricks/LTX-Video.")
        config = json.loads(json.dumps(NATIVE_VAE_CONFIG))
        sd, _ = _read_safetensors(os.path.join(path, "vae", "diffusion_pytorch_model.safetensors"), device)
        sd = _rename(sd, VAE_KEYS_RENAME)
    elif path.endswith(".safetensors"):
        sd, meta = _read_safetensors(path, device)
        config = json.loads(meta["config"])["vae"]
    else:
        raise ValueError(f"unrecognised checkpoint path: {path}")
    config["build_encoder"] = bool(with_encoder)
    vae = CausalVideoAutoencoder.from_config(config)
    if any(k.startswith("vae.") for k in sd):
        sd = {k.replace("vae.", "", 1): v for k, v in sd.items() if k.startswith("vae.")}
    want = vae.state_dict()
    missing = [k for k in want if k not in sd]
    if missing:
        raise KeyError(f"checkpoint is missing VAE tensors: {missing[:5]} ...")
    vae.load_state_dict({k: v for k, v in sd.items() if k in want}, strict=True)
    return vae.to(device=device, dtype=dtype).eval()
