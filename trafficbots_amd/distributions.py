"""Latent / destination distributions handed across the reference's model interface.

Mirrors the surface of `src/models/modules/distributions.py` (`MyDist.sample / log_prob`, `DiagGaussian`, `DestCategorical`,
`repeat_interleave_`) for the two distributions the default config uses.  The arithmetic -- mean + std * eps with the per-agent
deterministic blend, the Gaussian log-prob, log-softmax / arg max / inverse-CDF draw / log-prob gather over the polylines -- runs in the
HIP library (`tb_latent_sample`, `tb_dest_sample`, csrc/tb_sample_kernels.hip; the rollout prologue draws the personalities the same
way).  An object is served by the engine it is bound to: the model binds what it returns (`latent_encoder()`, `pred_goal()`) and what
it is given (`init(latent, ...)`); one made by hand takes `engine=`.  There is no host fallback: `sample` / `log_prob` on an unbound
object raise.

Unlike the reference, random draws are explicit: `sample(deterministic, eps=...)` takes the standard-normal draws (goldens pass the
reference's), `DestCategorical.sample` takes uniform draws `u` in [0, 1) (or a `torch.Generator` to make them) -- the reference calls
`torch.multinomial` on torch's global stream, which no other implementation can replay.

`repeat_interleave_(k, 0)` does not copy anything K times: the object remembers K and the kernels read scene n // K.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
from torch import Tensor


def _unbound(what: str):
    return RuntimeError(f"{what}: not bound to a device engine (pass engine=wm.engine, or hand the object to the model: "
                        "TrafficBots.init / latent_encoder / pred_goal bind it); there is no host fallback")


def _draw(fn, shape, device, generator) -> Tensor:
    """Random numbers from torch's generators (random-number generation is not arithmetic of the path); a generator of another device
    draws there and the numbers are moved."""
    if generator is not None and generator.device.type != torch.device(device).type:
        return fn(shape, device=generator.device, dtype=torch.float32, generator=generator).to(device)
    return fn(shape, device=device, dtype=torch.float32, generator=generator)


class DiagGaussian:
    """Independent Normal over the last dim (`distributions.py:40-59`).  `mean` [B, A, D]; `log_std` [D] (or broadcastable to it)."""

    def __init__(self, mean: Tensor, log_std: Tensor, valid: Optional[Tensor] = None, engine=None) -> None:
        self._mean_scene = mean
        self.log_std = log_std
        self._valid_scene = valid
        self._k = 1
        self.engine = engine

    # the reference's attributes, at the repeated size (views / copies of data, no arithmetic)
    @property
    def mean(self) -> Tensor:
        return self._mean_scene if self._k == 1 else self._mean_scene.repeat_interleave(self._k, 0)

    @property
    def valid(self) -> Optional[Tensor]:
        v = self._valid_scene
        return v if v is None or self._k == 1 else v.repeat_interleave(self._k, 0)

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        assert dim == 0, "the distributions are repeated over the batch axis only (waymo_motion.py:493)"
        self._k *= int(repeats)

    def _draws(self, deterministic: Union[bool, Tensor], eps: Optional[Tensor], generator):
        """(eps, deterministic tensor) as the kernels take them: eps None = every agent takes the mean."""
        if isinstance(deterministic, bool):
            if deterministic:
                return None, None
            det = None
        else:
            det = deterministic
        if eps is None:
            n = self._mean_scene.shape[0] * self._k
            eps = _draw(torch.randn, (n,) + tuple(self._mean_scene.shape[1:]), self._mean_scene.device, generator)
        return eps, det

    def sample(self, deterministic: Union[bool, Tensor], eps: Optional[Tensor] = None, generator=None) -> Tensor:
        """`MyDist.sample` (`distributions.py:18-38`): mean where deterministic, mean + std * eps elsewhere."""
        if self.engine is None:
            raise _unbound("DiagGaussian.sample")
        eps, det = self._draws(deterministic, eps, generator)
        z, _ = self.engine.latent_sample(self._mean_scene, self._k, eps=eps, deterministic=det, log_std=self.log_std, want_log_prob=False)
        return z

    def log_prob(self, sample: Tensor) -> Tensor:
        if self.engine is None:
            raise _unbound("DiagGaussian.log_prob")
        _, lp = self.engine.latent_sample(self._mean_scene, self._k, forced=sample, log_std=self.log_std, want_sample=False)
        return lp


class DestCategorical:
    """Categorical over map polylines (`distributions.py:158-201`) from the masked, un-normalised logits [B, A, P]."""

    def __init__(self, logits: Tensor, valid: Optional[Tensor] = None, engine=None):
        self._logits_scene = logits
        self._valid_scene = valid
        self._k = 1
        self._from_probs = False  # after repeat_interleave_ the reference holds Categorical(probs=softmax) (:196-199)
        self.engine = engine

    @property
    def valid(self) -> Optional[Tensor]:
        v = self._valid_scene
        return v if v is None or self._k == 1 else v.repeat_interleave(self._k, 0)

    @property
    def probs(self) -> Tensor:
        """softmax of the logits, [B * K, A, P] (`DestCategorical.probs`)."""
        if self.engine is None:
            raise _unbound("DestCategorical.probs")
        p = self.engine.dest_sample(self._logits_scene, 1, want_probs=True)[2]
        return p if self._k == 1 else p.repeat_interleave(self._k, 0)

    def repeat_interleave_(self, repeats: int, dim: int) -> None:
        assert dim == 0
        self._k *= int(repeats)
        self._from_probs = True

    def log_prob(self, sample: Tensor) -> Tensor:
        if self.engine is None:
            raise _unbound("DestCategorical.log_prob")
        return self.engine.dest_sample(self._logits_scene, self._k, forced=sample, from_probs=self._from_probs)[1]

    def sample(self, deterministic: Union[bool, Tensor], u: Optional[Tensor] = None, generator=None) -> Tensor:
        """Arg max of the probabilities where deterministic, the inverse CDF of the uniform draws `u` [N, A] elsewhere
        (`DestCategorical.sample`, `distributions.py:178-193`: the reference draws with torch.multinomial)."""
        if self.engine is None:
            raise _unbound("DestCategorical.sample")
        det = None
        if isinstance(deterministic, bool):
            if deterministic:
                u = None
            elif u is None:
                u = self._uniform(generator)
        else:
            det = deterministic
            if u is None:
                u = self._uniform(generator)
        return self.engine.dest_sample(self._logits_scene, self._k, uniform=u, deterministic=det, from_probs=self._from_probs)[0].long()

    def _uniform(self, generator) -> Tensor:
        b, a, _ = self._logits_scene.shape
        return _draw(torch.rand, (b * self._k, a), self._logits_scene.device, generator)
