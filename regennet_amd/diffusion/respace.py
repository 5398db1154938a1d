"""Timestep respacing — mirror of the reference's `diffusion/respace.py` (space_timesteps :8-61,
SpacedDiffusion :64-121, _WrappedModel :124-129). Integer results are bit-exact by construction
(pure Python integer/`round` arithmetic, checked against golden vectors recorded from the reference)."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Kept-timestep set. "ddimN": first integer stride giving exactly N points; "a,b,c": per-section
    rounded fractional strides (respace.py:30-60)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                picked = range(0, num_timesteps, stride)
                if len(picked) == desired:
                    return set(picked)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, offset = [], 0
    for idx, count in enumerate(section_counts):
        size = base + (1 if idx < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(offset + round(pos))
            pos += stride
        offset += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion that keeps only `use_timesteps` of a base process (respace.py:64-87): betas are re-derived
    from the base cumulative alphas so that the marginals at kept steps are unchanged."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        base_betas = np.array(kwargs["betas"], dtype=np.float64)
        self.original_num_steps = len(base_betas)
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        tmap, new_betas, last = [], [], 1.0
        for i, a in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - a / last)
                last = a
                tmap.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)
        self.timestep_map = tmap

    def _model_timesteps(self, t):
        # _WrappedModel.__call__ (respace.py:124-129): integer gather, same dtype as t
        m = th.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]
        return m.float() * (1000.0 / self.original_num_steps) if self.rescale_timesteps else m

    def _wrap_model(self, model):
        return model if isinstance(model, _WrappedModel) else _WrappedModel(
            model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        new_ts = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
