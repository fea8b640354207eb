"""Import shims that let the reference's OWN hot-path modules be imported, untouched,
from /root/reference inside the build container.

BUILD-CONTAINER-ONLY TOOLING: used by oracle/gen/make_golden.py to produce the
fixtures in tests/golden/.  Nothing here ships in the product path and nothing here
is used on the GPU box (where /root/reference does not exist).

What is shimmed (all absent from this image, none installable -- no network):
  * ``mmgp``           -> only ``offload.shared_state = {"_attention": "sdpa"}`` (the
                          reference's eager backend selector, wan/modules/attention.py:183)
  * ``wan``/``wan.modules`` -> empty namespace packages whose ``__path__`` points into
                          /root/reference/wan, so ``wan/__init__.py`` (which drags in
                          easydict/torchvision) is skipped while
                          ``wan.modules.attention`` itself is the reference's file
  * ``torch.cuda.get_device_capability`` -> (9, 4) (called at import,
                          wan/modules/attention.py:7; there is no GPU here)
  * ``diffusers``      -> the mixins as inert bases, and the handful of leaf modules
                          the hot path instantiates, implemented on top of
                          ``oracle.leaves`` (published definitions; PARITY UNPINNED
                          at that boundary, see oracle/__init__.py)
"""
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self._internal_config = _Cfg(cfg)
    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return getattr(self, "_internal_config", _Cfg())

    @classmethod
    def from_config(cls, config, **kw):
        config = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        return cls(**config)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class SchedulerMixin:
    pass


class BaseOutput(dict):
    def __post_init__(self):
        for k, v in self.__dict__.items():
            self[k] = v


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Dummy(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


# ---- leaf modules (diffusers published definitions, arithmetic in oracle.leaves) ----
def _leaves():
    from oracle import leaves
    return leaves


class RMSNorm(nn.Module):
    def __init__(self, dim, eps, elementwise_affine=True):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None

    def forward(self, x):
        return _leaves().rms_norm(x, self.eps, self.weight)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class PixArtAlphaCombinedTimestepSizeEmbeddings(nn.Module):
    def __init__(self, embedding_dim, size_emb_dim, use_additional_conditions=False):
        super().__init__()
        assert not use_additional_conditions
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)

    def forward(self, timestep, resolution, aspect_ratio, batch_size, hidden_dtype):
        # the sinusoid is the REFERENCE's own in-repo copy (ltx_video/models/transformers/embeddings.py:10-50, imports
        # untouched), not the oracle's restatement: the goldens do not depend on oracle code here (G0 pins the two)
        from ltx_video.models.transformers.embeddings import get_timestep_embedding
        proj = get_timestep_embedding(timestep, 256, flip_sin_to_cos=True, downscale_freq_shift=0.0)
        return self.timestep_embedder(proj.to(dtype=hidden_dtype))


class AdaLayerNormSingle(nn.Module):
    def __init__(self, embedding_dim, use_additional_conditions=False):
        super().__init__()
        self.emb = PixArtAlphaCombinedTimestepSizeEmbeddings(embedding_dim, embedding_dim // 3,
                                                             use_additional_conditions)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=True)

    def forward(self, timestep, added_cond_kwargs=None, batch_size=None, hidden_dtype=None):
        emb = self.emb(timestep, **added_cond_kwargs, batch_size=batch_size, hidden_dtype=hidden_dtype)
        return self.linear(self.silu(emb)), emb


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, out_features or hidden_size)

    def forward(self, caption):
        return self.linear_2(F.gelu(self.linear_1(caption), approximate="tanh"))


class DecoderOutput(BaseOutput):
    def __init__(self, sample):
        super().__init__(sample=sample)
        self.sample = sample


def install():
    """Register every shim in sys.modules and put the reference on sys.path."""
    if "mmgp" in sys.modules:
        return
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    torch.cuda.get_device_capability = lambda *a, **k: (9, 4)

    off = types.SimpleNamespace(shared_state={"_attention": "sdpa"})
    _mod("mmgp", offload=off)

    wan = _mod("wan")
    wan.__path__ = [REFERENCE_ROOT + "/wan"]
    wm = _mod("wan.modules")
    wm.__path__ = [REFERENCE_ROOT + "/wan/modules"]

    d = _mod("diffusers", ConfigMixin=ConfigMixin, ModelMixin=ModelMixin, AutoencoderKL=_Dummy)
    d.__path__ = []
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    m = _mod("diffusers.models", AutoencoderKL=_Dummy)
    m.__path__ = []
    _mod("diffusers.models.embeddings", PixArtAlphaTextProjection=PixArtAlphaTextProjection,
         PixArtAlphaCombinedTimestepSizeEmbeddings=PixArtAlphaCombinedTimestepSizeEmbeddings)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.normalization", AdaLayerNormSingle=AdaLayerNormSingle, RMSNorm=RMSNorm)
    _mod("diffusers.models.activations", GEGLU=_Dummy, GELU=GELU, ApproximateGELU=_Dummy)
    _mod("diffusers.models.attention", _chunked_feed_forward=None)
    _mod("diffusers.models.attention_processor", LoRAAttnAddedKVProcessor=_Dummy,
         LoRAAttnProcessor=_Dummy, LoRAAttnProcessor2_0=_Dummy, LoRAXFormersAttnProcessor=_Dummy,
         SpatialNorm=_Dummy)
    _mod("diffusers.models.lora", LoRACompatibleLinear=_Dummy)
    ae = _mod("diffusers.models.autoencoders")
    ae.__path__ = []
    _mod("diffusers.models.autoencoders.vae", DecoderOutput=DecoderOutput,
         DiagonalGaussianDistribution=_Dummy)
    _mod("diffusers.models.modeling_outputs", AutoencoderKLOutput=_Dummy)
    sch = _mod("diffusers.schedulers")
    sch.__path__ = []
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin)
    u = _mod("diffusers.utils", BaseOutput=BaseOutput, is_torch_version=lambda *a: True,
             logging=types.SimpleNamespace(get_logger=lambda *a, **k: _Logger()),
             deprecate=lambda *a, **k: None)
    u.__path__ = []
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c,
         randn_tensor=None)


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor, single-generator case: torch.randn on the
    generator's device (restated leaf, PARITY UNPINNED like the other diffusers leaves)."""
    return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)


class VaeImageProcessor:
    """diffusers.image_processor.VaeImageProcessor, the part the pipeline calls (pipeline_ltx_video.py:303, 1299):
    ``postprocess(image, output_type)`` with do_normalize=True -- "latent" passes through, "pt" de-normalises
    ``(x / 2 + 0.5).clamp(0, 1)`` (restated leaf, PARITY UNPINNED like the other diffusers leaves)."""

    def __init__(self, vae_scale_factor=8, **kw):
        self.vae_scale_factor = vae_scale_factor

    @staticmethod
    def postprocess(image, output_type="pil", do_denormalize=None):
        if output_type == "latent":
            return image
        if output_type != "pt":
            raise ValueError("only 'pt' / 'latent' are restated here (video tensors)")
        return torch.stack([(image[i] / 2 + 0.5).clamp(0, 1) for i in range(image.shape[0])])


def install_pipeline_leaves():
    """Extra inert leaves so that ``ltx_video.pipelines.pipeline_ltx_video`` imports: its
    static/helper methods (prepare_conditioning, denoising_step, add_noise_to_image_conditioning_latents,
    the latent upsampler bridge ...) are then callable with a light stand-in ``self``."""
    install()
    _mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    p = _mod("diffusers.pipelines")
    p.__path__ = []
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=type("DiffusionPipeline", (), {}),
         ImagePipelineOutput=_Dummy)
    sys.modules["diffusers.schedulers"].DPMSolverMultistepScheduler = _Dummy
    sys.modules["diffusers.utils.torch_utils"].randn_tensor = randn_tensor
