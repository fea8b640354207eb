"""``RectifiedFlowScheduler`` -- drop-in for ltx_video/schedulers/rf.py:176-392 (Uniform sampler,
SD3 resolution-dependent shift).  The schedule itself is a handful of host scalars per
generation; the per-step Euler update ``x - dt * v`` (rf.py:375) runs inside
``ltxmi_guidance_step_bf16`` on the loop's fast path (see pipeline.py) and as plain tensor
arithmetic in ``step`` for callers that use the scheduler directly."""
import math
from dataclasses import dataclass
from typing import Optional

import torch


def time_shift(mu: float, sigma: float, t):                                   # rf.py:69-70
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def get_normal_shift(n_tokens, min_tokens=1024, max_tokens=4096, min_shift=0.95, max_shift=2.05):   # rf.py:73-82
    m = (max_shift - min_shift) / (max_tokens - min_tokens)
    return m * n_tokens + (min_shift - m * min_tokens)


def strech_shifts_to_terminal(shifts, terminal=0.1):                          # rf.py:85-109
    if shifts.numel() == 0:
        raise ValueError("The 'shifts' tensor must not be empty.")
    if terminal <= 0 or terminal >= 1:
        raise ValueError("The terminal value must be between 0 and 1 (exclusive).")
    one_minus_z = 1 - shifts
    return 1 - (one_minus_z / (one_minus_z[-1] / (1 - terminal)))


def sd3_resolution_dependent_timestep_shift(samples_shape, timesteps, target_shift_terminal=None):  # rf.py:112-149
    if len(samples_shape) == 3:
        _, m, _ = samples_shape
    elif len(samples_shape) in [4, 5]:
        m = math.prod(samples_shape[2:])
    else:
        raise ValueError("Samples must have shape (b, t, c), (b, c, h, w) or (b, c, f, h, w)")
    ts = time_shift(get_normal_shift(m), 1, timesteps)
    if target_shift_terminal is not None:
        ts = strech_shifts_to_terminal(ts, target_shift_terminal)
    return ts


@dataclass
class RectifiedFlowSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class RectifiedFlowScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, shifting: Optional[str] = None, base_resolution=32 ** 2,
                 target_shift_terminal: Optional[float] = None, sampler: Optional[str] = "Uniform",
                 shift: Optional[float] = None, **ignored):
        if sampler != "Uniform":
            raise NotImplementedError("only the 'Uniform' sampler is on this path")
        if shifting not in (None, "SD3"):
            raise NotImplementedError("only SD3 shifting is on this path")
        self.num_train_timesteps = num_train_timesteps
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.shifting = shifting
        self.target_shift_terminal = target_shift_terminal
        self.timesteps = self.sigmas = torch.linspace(1, 1 / num_train_timesteps, num_train_timesteps)

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    def shift_timesteps(self, samples_shape, timesteps):
        if self.shifting == "SD3":
            return sd3_resolution_dependent_timestep_shift(samples_shape, timesteps, self.target_shift_terminal)
        return timesteps

    def set_timesteps(self, num_inference_steps=None, samples_shape=None, timesteps=None, device=None):   # rf.py:227-261
        if timesteps is not None and num_inference_steps is not None:
            raise ValueError("You cannot provide both `timesteps` and `num_inference_steps`.")
        if timesteps is None:
            num_inference_steps = min(self.num_train_timesteps, num_inference_steps)
            timesteps = torch.linspace(1, 1 / num_inference_steps, num_inference_steps)
            timesteps = self.shift_timesteps(samples_shape, timesteps)
        else:
            timesteps = torch.Tensor(timesteps)
            num_inference_steps = len(timesteps)
        self.host_timesteps = [float(x) for x in timesteps.tolist()]      # host copy: no per-step D2H
        self.timesteps = timesteps.to(device) if device is not None else timesteps
        self.num_inference_steps = num_inference_steps
        self.sigmas = self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def host_dt(self, t: float, t_eps: float = 1e-6) -> float:
        """dt to the closest lower timestep for a global (scalar) timestep, rf.py:355-359."""
        lower = [x for x in self.host_timesteps + [0.0] if x < t - t_eps]
        return t - lower[0]

    def add_noise(self, original_samples, noise, timesteps):                         # rf.py:382-392
        sigmas = timesteps
        while sigmas.ndim < original_samples.ndim:
            sigmas = sigmas[..., None]
        return (1 - sigmas) * original_samples + sigmas * noise

    def step(self, model_output, timestep, sample, return_dict=True, stochastic_sampling=False, **kwargs):   # rf.py:311-380
        """``generator`` (extension): the RNG of the stochastic branch's noise (the reference uses the global one)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t_eps = 1e-6
        padded = torch.cat([self.timesteps, torch.zeros(1, device=self.timesteps.device)])
        if timestep.ndim == 0:
            lower = padded[padded < timestep - t_eps][0]
            dt = timestep - lower
        else:
            assert timestep.ndim == 2
            mask = padded[:, None, None] < timestep[None] - t_eps
            lower, _ = (mask * padded[:, None, None]).max(dim=0)
            dt = (timestep - lower)[..., None]
        if stochastic_sampling:                                                       # rf.py:368-373
            x0 = sample - timestep[..., None] * model_output
            noise = torch.randn(sample.shape, device=sample.device, dtype=sample.dtype, generator=kwargs.get("generator"))
            prev = self.add_noise(x0, noise, timestep[..., None] - dt)
        else:
            prev = sample - dt * model_output
        if not return_dict:
            return (prev,)
        return RectifiedFlowSchedulerOutput(prev_sample=prev)
