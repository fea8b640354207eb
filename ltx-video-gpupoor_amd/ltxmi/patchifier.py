"""``SymmetricPatchifier`` -- drop-in for ltx_video/models/transformers/symmetric_patchifier.py
(layout only: b c f h w <-> b (f h w) c for patch size 1) plus ``latent_to_pixel_coords``
(ltx_video/models/autoencoders/vae_encode.py:190-225)."""
import torch


class SymmetricPatchifier:
    def __init__(self, patch_size: int = 1):
        if patch_size != 1:
            raise NotImplementedError("LTX-Video uses patch_size 1 on the DiT side")
        self._patch_size = (1, patch_size, patch_size)

    @property
    def patch_size(self):
        return self._patch_size

    def get_latent_coords(self, latent_num_frames, latent_height, latent_width, batch_size, device):   # :33-51
        grid = torch.meshgrid(torch.arange(0, latent_num_frames, device=device),
                              torch.arange(0, latent_height, device=device),
                              torch.arange(0, latent_width, device=device), indexing="ij")
        coords = torch.stack(grid, dim=0).unsqueeze(0).repeat(batch_size, 1, 1, 1, 1)
        return coords.reshape(batch_size, 3, -1)

    def patchify(self, latents):                                             # :55-65
        b, c, f, h, w = latents.shape
        coords = self.get_latent_coords(f, h, w, b, latents.device)
        return latents.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c), coords

    def unpatchify(self, latents, output_height, output_width, out_channels):  # :67-84
        b, n, c = latents.shape
        f = n // (output_height * output_width)
        return latents.reshape(b, f, output_height, output_width, c).permute(0, 4, 1, 2, 3)


def latent_to_pixel_coords_from_factors(latent_coords, scale_factors, causal_fix=False):   # vae_encode.py:214-225
    pc = latent_coords * torch.tensor(scale_factors, device=latent_coords.device)[None, :, None]
    if causal_fix:
        pc[:, 0] = (pc[:, 0] + 1 - scale_factors[0]).clamp(min=0)
    return pc
