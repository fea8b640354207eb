"""ltxmi -- MI355X-native LTX-Video denoise hot path (DiT forward + causal 3-D VAE decode).

Host-side mirror of the reference's operator interface for this path; every kernel lives in
``libltxmi.so`` (hand-written HIP for gfx950, C ABI in include/ltxmi.h).  Importing this
package loads the library and raises if it is missing: there is no fallback path.
"""
from . import _lib  # noqa: F401  (fails loudly when libltxmi.so is absent)
from .attention import (Attention, AttnProcessor2_0, BasicTransformerBlock, FeedForward,  # noqa: F401
                        SkipLayerStrategy)
from .attention_seam import pay_attention  # noqa: F401
from .autoencoder import (CausalVideoAutoencoder, DecoderOutput, AutoencoderKLOutput,  # noqa: F401
                          DiagonalGaussianDistribution, normalize_latents, un_normalize_latents, vae_decode, vae_encode)
from .patchifier import SymmetricPatchifier, latent_to_pixel_coords_from_factors  # noqa: F401
from .pipeline import ConditioningItem, LTXMultiScalePipeline, LTXVideoPipeline, retrieve_timesteps  # noqa: F401
from .latent_upsampler import LatentUpsampler, adain_filter_latent, upsample_latents  # noqa: F401
from .scheduler import RectifiedFlowScheduler  # noqa: F401
from .transformer3d import Transformer3DModel, Transformer3DModelOutput  # noqa: F401

__version__ = _lib.lib.ltxmi_version().decode()
from .ops import set_step_invariant_caching  # noqa: E402,F401
