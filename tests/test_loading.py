"""Checkpoint formats of the reference (single-file safetensors with a config blob; diffusers
directory layout with renamed keys) load into the drop-in modules.  CPU only (module construction and
state-dict plumbing; no kernels run)."""
import json
import os

import torch
from safetensors.torch import save_file

from oracle import dit, vae as ov


def test_transformer_single_file_with_config_metadata(tmp_path):
    from ltxmi import Transformer3DModel
    cfg = dict(dit.default_2b_config(), num_attention_heads=2, attention_head_dim=64, num_layers=2,
               cross_attention_dim=128, caption_channels=128)
    sd = dit.init_state_dict(cfg, seed=0)
    blob = {"model.diffusion_model." + k: v.to(torch.bfloat16) for k, v in sd.items()}
    blob["vae.decoder.conv_in.conv.weight"] = torch.zeros(1)          # other sections share the file
    path = os.path.join(tmp_path, "ckpt.safetensors")
    save_file(blob, path, metadata={"config": json.dumps({"transformer": cfg, "vae": {}})})
    m = Transformer3DModel.from_pretrained(path, device="cpu")
    assert m.dtype == torch.bfloat16 and len(m.transformer_blocks) == 2
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k].to(torch.bfloat16)), k


def test_transformer_diffusers_directory_renames_keys(tmp_path):
    from ltxmi import loading
    # a 1-layer stand-in for the file contents; the config check itself must accept only the published one
    os.makedirs(os.path.join(tmp_path, "transformer"))
    json.dump({"bogus": 1}, open(os.path.join(tmp_path, "transformer", "config.json"), "w"))
    try:
        loading.load_transformer(str(tmp_path), device="cpu")
        raise AssertionError("an unknown diffusers config must be rejected")
    except ValueError as e:
        assert "not supported" in str(e)
    renamed = loading._rename({"proj_in.weight": 0, "time_embed.linear.bias": 1,
                               "transformer_blocks.0.attn1.norm_q.weight": 2}, loading.TRANSFORMER_KEYS_RENAME)
    assert set(renamed) == {"patchify_proj.weight", "adaln_single.linear.bias", "transformer_blocks.0.attn1.q_norm.weight"}
    assert loading.NATIVE_2B_TRANSFORMER_CONFIG["num_layers"] == 28


def test_vae_single_file_and_diffusers_key_map(tmp_path):
    from ltxmi import CausalVideoAutoencoder, loading
    cfg = ov.demo_config(128)
    cfg["decoder_base_channels"] = 64
    sd = ov.init_state_dict(cfg, seed=1)
    blob = {"vae." + k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd.items()}
    blob["vae.encoder.conv_in.conv.weight"] = torch.zeros(1)          # encoder tensors are ignored (decode side)
    path = os.path.join(tmp_path, "vae.safetensors")
    save_file(blob, path, metadata={"config": json.dumps({"vae": cfg})})
    v = CausalVideoAutoencoder.from_pretrained(path, device="cpu")
    got = v.state_dict()
    for k in sd:
        assert k in got and got[k].shape == sd[k].shape, k
    # diffusers -> native decoder block names
    r = loading._rename({"decoder.mid_block.resnets.0.conv1.conv.weight": 0,
                         "decoder.up_blocks.1.upsamplers.0.conv.conv.weight": 1,
                         "latents_std": 2}, loading.VAE_KEYS_RENAME)
    assert set(r) == {"decoder.up_blocks.0.res_blocks.0.conv1.conv.weight", "decoder.up_blocks.2.conv.conv.weight",
                      "per_channel_statistics.std-of-means"}


def test_vae_single_file_with_encoder(tmp_path):
    """Encoder tensors load under the reference's key names (encoder.down_blocks.N....)."""
    from ltxmi import CausalVideoAutoencoder
    from oracle import vae_encoder as oe
    cfg = ov.demo_config(128)
    cfg["decoder_base_channels"] = 64
    cfg["encoder_blocks"] = oe.demo_encoder_blocks()
    cfg["encoder_base_channels"] = 64
    sd = dict(ov.init_state_dict(cfg, seed=1))
    sd.update(oe.init_state_dict(cfg, seed=2))
    blob = {"vae." + k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd.items()}
    path = os.path.join(tmp_path, "vae.safetensors")
    save_file(blob, path, metadata={"config": json.dumps({"vae": cfg})})
    v = CausalVideoAutoencoder.from_pretrained(path, device="cpu")
    got = v.state_dict()
    assert any(k.startswith("encoder.down_blocks.1.conv.conv.") for k in got)
    for k in sd:
        assert k in got and got[k].shape == sd[k].shape, k
    assert set(got) == set(sd)


def test_packed_weights_follow_in_place_edits():
    """The product-side packed copies ([to_q; to_k; to_v], [to_k; to_v], the tap-major conv weights) key on storage AND
    version of their sources: an in-place edit (what a LoRA merge does) must rebuild them, and a self-attention module
    never builds the cross-attention pack (nor the reverse)."""
    import torch
    from ltxmi.attention import Attention
    from ltxmi.autoencoder import CausalConv3d
    att = Attention(query_dim=128, heads=2, dim_head=64, bias=True, qk_norm="rms_norm")
    w1, b1 = att.packed_qkv()
    assert w1.shape == (384, 128) and torch.equal(w1[:128], att.to_q.weight) and set(att._packs) == {"qkv"}
    assert att.packed_qkv()[0] is w1                                  # cached while nothing changes
    with torch.no_grad():
        att.to_k.weight.add_(1.0)                                     # in place: same storage, new version
    w2, _ = att.packed_qkv()
    assert w2 is not w1 and torch.equal(w2[128:256], att.to_k.weight)
    with torch.no_grad():
        att.to_v.bias.mul_(2.0)
    assert torch.equal(att.packed_qkv()[1][256:], att.to_v.bias)
    cross = Attention(query_dim=128, cross_attention_dim=128, heads=2, dim_head=64, bias=True, qk_norm="rms_norm")
    wkv, _ = cross.packed_kv()
    assert wkv.shape == (256, 128) and set(cross._packs) == {"kv"}
    conv = CausalConv3d(64, 64)
    p1, _ = conv.packed()
    with torch.no_grad():
        conv.conv.weight.mul_(0.5)
    p2, _ = conv.packed()
    assert p2 is not p1 and torch.allclose(p2.float(), p1.float() * 0.5, atol=1e-2)
    # what the key can NOT see: an edit through .data bumps no version counter -> the pack is stale until
    # invalidate_packed() (documented on Attention._pack / CausalConv3d.packed)
    v0 = att.to_q.weight._version
    att.to_q.weight.data.add_(1.0)
    assert att.to_q.weight._version == v0
    stale, _ = att.packed_qkv()
    assert not torch.equal(stale[:128], att.to_q.weight)
    att.invalidate_packed()
    assert torch.equal(att.packed_qkv()[0][:128], att.to_q.weight)
    conv.conv.weight.data.mul_(2.0)
    conv.invalidate_packed()
    assert torch.allclose(conv.packed()[0].float(), p1.float(), atol=1e-2)
    # inference tensors track no version counter at all: the key must not raise on them (storage-only key)
    with torch.inference_mode():
        att_i = Attention(query_dim=128, heads=2, dim_head=64, bias=True, qk_norm="rms_norm")
        conv_i = CausalConv3d(64, 64)
    assert att_i.to_q.weight.is_inference()
    wi, _ = att_i.packed_qkv()
    assert wi.shape == (384, 128) and att_i.packed_qkv()[0] is wi
    assert conv_i.packed()[0].shape == (64, 27 * 64)
