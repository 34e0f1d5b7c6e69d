"""Latent / destination distributions handed across the reference's model interface.

Mirrors the surface of `src/models/modules/distributions.py` (`MyDist.sample/log_prob`,
`DiagGaussian`, `DestCategorical`, `repeat_interleave_`) for the two distributions the default
config uses.  Arithmetic here is a handful of element-wise tensor ops on tiny tensors
([N, A, 16] and [N, A, P]); the heavy work is in the HIP library.

Unlike the reference, random draws are explicit: `sample(deterministic, eps=...)` takes the
standard-normal draws (goldens pass the reference's), `DestCategorical.sample` takes a
`torch.Generator`.
"""
from __future__ import annotations

import math
from typing import Optional, Union

import torch
from torch import Tensor


class DiagGaussian:
    """Independent Normal over the last dim (`distributions.py:40-59`)."""

    def __init__(self, mean: Tensor, log_std: Tensor, valid: Optional[Tensor] = None) -> None:
        self.mean = mean
        self.log_std = log_std
        self.stddev = log_std.exp().expand_as(mean)
        self.valid = valid

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        self.mean = self.mean.repeat_interleave(repeats, dim)
        self.stddev = self.stddev.repeat_interleave(repeats, dim)
        if self.valid is not None:
            self.valid = self.valid.repeat_interleave(repeats, dim)

    def sample(self, deterministic: Union[bool, Tensor], eps: Optional[Tensor] = None, generator=None) -> Tensor:
        """`MyDist.sample` (`distributions.py:18-38`): mean where deterministic, mean + std*eps elsewhere."""
        if isinstance(deterministic, bool) and deterministic:
            return self.mean
        if eps is None:
            eps = torch.randn(self.mean.shape, device=self.mean.device, dtype=self.mean.dtype, generator=generator)
        rnd = self.mean + eps.to(self.mean.dtype) * self.stddev
        if isinstance(deterministic, bool):
            return rnd
        det = deterministic.unsqueeze(-1)
        return self.mean.masked_fill(~det, 0) + rnd.masked_fill(det, 0)

    def log_prob(self, sample: Tensor) -> Tensor:
        var = self.stddev ** 2
        lp = -((sample - self.mean) ** 2) / (2 * var) - self.stddev.log() - math.log(math.sqrt(2 * math.pi))
        return lp.sum(-1)


class DestCategorical:
    """Categorical over map polylines (`distributions.py:158-201`)."""

    def __init__(self, logits: Optional[Tensor] = None, probs: Optional[Tensor] = None, valid: Optional[Tensor] = None):
        if probs is None:
            assert logits is not None
            self.logits = logits - logits.logsumexp(-1, keepdim=True)
            self.probs = torch.softmax(logits, -1)
        else:
            self._set_probs(probs)
        self.valid = valid

    def _set_probs(self, probs: Tensor) -> None:
        # torch.distributions.Categorical(probs=...) normalises and clamps before the log
        self.probs = probs / probs.sum(-1, keepdim=True)
        eps = torch.finfo(self.probs.dtype).eps
        self.logits = torch.log(self.probs.clamp(min=eps, max=1 - eps))

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        self._set_probs(self.probs.repeat_interleave(repeats, dim))
        if self.valid is not None:
            self.valid = self.valid.repeat_interleave(repeats, dim)

    def log_prob(self, sample: Tensor) -> Tensor:
        return self.logits.gather(-1, sample.long().unsqueeze(-1)).squeeze(-1)

    def sample(self, deterministic: Union[bool, Tensor], generator=None) -> Tensor:
        det = self.probs.argmax(-1)
        if isinstance(deterministic, bool) and deterministic:
            return det
        flat = self.probs.reshape(-1, self.probs.shape[-1])
        rnd = torch.multinomial(flat, 1, generator=generator).view(self.probs.shape[:-1])
        if isinstance(deterministic, bool):
            return rnd
        return det.masked_fill(~deterministic, 0) + rnd.masked_fill(deterministic, 0)
