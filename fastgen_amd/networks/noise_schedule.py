"""Host-side noise schedules exposed as `net.noise_scheduler`: EDM ('edm') and rectified flow ('rf').

Mirrors the surface of the reference's `BaseNoiseSchedule` / `EDMNoiseSchedule` / `RFNoiseSchedule`
(fastgen/networks/noise_schedule.py:23-726, 729-1035, 1306-1486) that the sampling callers read
(methods/model.py:361-413, methods/consistency_model/mean_flow.py:336-381): get_t_list, latents, forward_process,
x0_to_eps, convert_model_output, max_t, max_sigma, t_precision, sigmas, is_t_valid, sample_t.  These are tiny tensor
expressions evaluated with torch on whatever device the inputs live on; inside the fused sampler (fg_sampler_run)
the same formulas run as HIP kernels (csrc/misc.hip) and these classes only supply the timestep list.
"""
from __future__ import annotations

from typing import Optional

import torch

NET_PRED_TYPES = {"x0", "eps", "v", "flow"}

_PRECISION = {"float64": torch.float64, "float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def expand_like(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Right-pad x with singleton dims up to target.ndim (fastgen/utils/__init__.py:22-50)."""
    x = torch.atleast_1d(x)
    return x.reshape(x.shape + (1,) * (target.ndim - x.ndim))


class BaseNoiseSchedule(torch.nn.Module):
    """x_t = alpha(t) x_0 + sigma(t) eps (noise_schedule.py:23-726).  Subclasses supply alpha and the sigma table;
    sigma(t) = t for both schedules on this path."""

    schedule_id = -1  # FG_SCHEDULE_* understood by fg_sampler_run

    def __init__(self, min_t: float, max_t: float, num_steps: int, clamp_min: float = 1e-6,
                 t_precision: str = "float64", **kwargs):
        super().__init__()
        self._min_t, self._max_t = float(min_t), float(max_t)
        self.num_steps = num_steps
        self.clamp_min = clamp_min
        self.t_precision = _PRECISION[t_precision]

    # -- scalars ---------------------------------------------------------------------------------------
    @property
    def min_t(self) -> float:
        return self._min_t

    @property
    def max_t(self) -> float:
        return self._max_t

    @property
    def sigmas(self) -> torch.Tensor:
        return self._sigmas

    def alpha(self, t):
        raise NotImplementedError

    def sigma(self, t):
        return t

    # -- helpers ---------------------------------------------------------------------------------------
    def is_t_valid(self, t: torch.Tensor) -> torch.Tensor:
        """min_t <= t <= max_t up to one ulp of t's dtype (noise_schedule.py:409-423)."""
        dt = t.dtype if t.dtype in (torch.bfloat16, torch.float32, torch.float64) else torch.float32
        lo = torch.nextafter(torch.tensor(self.min_t, dtype=dt, device=t.device), torch.tensor(-float("inf"), dtype=dt, device=t.device))
        hi = torch.nextafter(torch.tensor(self.max_t, dtype=dt, device=t.device), torch.tensor(float("inf"), dtype=dt, device=t.device))
        return torch.all((lo <= t) & (t <= hi))

    def non_zero_clamp(self, x: torch.Tensor) -> torch.Tensor:
        return torch.where(x >= 0, x.clamp(min=self.clamp_min), x.clamp(max=-self.clamp_min))

    # -- what the samplers call ------------------------------------------------------------------------------
    def get_t_list(self, sample_steps: int, device: Optional[torch.device] = None) -> torch.Tensor:
        """linspace(max_t, 0, sample_steps+1) (noise_schedule.py:259-272)."""
        t = torch.linspace(self.max_t, 0, sample_steps + 1, device=device or self._sigmas.device, dtype=self.t_precision)
        return t.clamp(max=self.max_t)

    def latents(self, noise: torch.Tensor, t_init: Optional[torch.Tensor] = None) -> torch.Tensor:
        """noise * sigma(t_init), evaluated in fp64 (noise_schedule.py:72-88)."""
        if t_init is None:
            t_init = torch.as_tensor(self.max_t, dtype=self.t_precision, device=noise.device)
        assert self.is_t_valid(t_init), f"t_init must be in [{self.min_t}, {self.max_t}], but got {t_init}"
        s = expand_like(self.sigma(t_init.to(torch.float64)), noise)
        return (noise.to(torch.float64) * s).to(noise.dtype)

    def forward_process(self, x: torch.Tensor, eps: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """alpha(t) x + sigma(t) eps in fp64, cast back (noise_schedule.py:425-449)."""
        assert self.is_t_valid(t), f"t must be in [{self.min_t}, {self.max_t}], but got {t}"
        t64 = t.to(torch.float64)
        out = x.to(torch.float64) * expand_like(self.alpha(t64), x) + eps.to(torch.float64) * expand_like(self.sigma(t64), eps)
        return out.to(x.dtype)

    def x0_to_eps(self, xt: torch.Tensor, x0: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """(x_t - alpha x_0) / clamp(sigma) in fp64 (noise_schedule.py:544-574)."""
        assert self.is_t_valid(t), f"t must be in [{self.min_t}, {self.max_t}], but got {t}"
        t64 = t.to(torch.float64)
        num = xt.to(torch.float64) - x0.to(torch.float64) * expand_like(self.alpha(t64), xt)
        return (num / self.non_zero_clamp(expand_like(self.sigma(t64), xt))).to(xt.dtype)

    def eps_to_x0(self, xt, eps, t):
        """(x_t - sigma eps) / clamp(alpha) in fp64 (noise_schedule.py:576-608)."""
        t64 = t.to(torch.float64)
        out = (xt.to(torch.float64) - eps.to(torch.float64) * expand_like(self.sigma(t64), xt)) / self.non_zero_clamp(
            expand_like(self.alpha(t64), xt))
        return out.to(xt.dtype)

    def x0_to_flow(self, xt, x0, t):
        """(x_t - x_0) / clamp(t): the EDM and RF overrides coincide (noise_schedule.py:1006-1035, 1457-1486)."""
        te = expand_like(t.to(torch.float64), xt)
        return ((xt.to(torch.float64) - x0.to(torch.float64)) / self.non_zero_clamp(te)).to(xt.dtype)

    def flow_to_x0(self, xt, v, t):
        """x_t - t v (noise_schedule.py:975-1004, 1426-1455)."""
        te = expand_like(t.to(torch.float64), xt)
        return (xt.to(torch.float64) - v.to(torch.float64) * te).to(xt.dtype)

    def convert_model_output(self, xt, model_output, t, src_pred_type: str = "x0", target_pred_type: str = "eps"):
        """Prediction-type conversion through x0 (noise_schedule.py:666-726); 'v' needs alpha^2+sigma^2=1 and is not
        defined for the EDM / RF schedules."""
        if src_pred_type == target_pred_type:
            return model_output
        for p in (src_pred_type, target_pred_type):
            if p not in NET_PRED_TYPES:
                raise ValueError(f"Unsupported pred type '{p}'. Supported types: {NET_PRED_TYPES}")
            if p == "v":
                raise AssertionError("Only AlphaNoiseSchedule supports v-prediction!")
        x0 = {"x0": lambda: model_output, "eps": lambda: self.eps_to_x0(xt, model_output, t),
              "flow": lambda: self.flow_to_x0(xt, model_output, t)}[src_pred_type]()
        return {"x0": lambda: x0, "eps": lambda: self.x0_to_eps(xt, x0, t),
                "flow": lambda: self.x0_to_flow(xt, x0, t)}[target_pred_type]()


class EDMNoiseSchedule(BaseNoiseSchedule):
    """x_t = x_0 + t * eps with t in [0.002, 80]: alpha(t) = 1, sigma(t) = t (noise_schedule.py:729-777)."""

    schedule_id = 0

    def __init__(self, min_t: float = 0.002, max_t: float = 80.0, rho: float = 7.0, min_step_percent: float = 0.002,
                 max_step_percent: float = 0.998, num_steps: int = 1000, **kwargs):
        super().__init__(min_t, max_t, num_steps, **kwargs)
        # Karras rho-schedule sampled on num_steps points, increasing (noise_schedule.py:752-756); plain attribute,
        # not a buffer: the reference's scheduler contributes no state-dict entries.
        ramp = torch.linspace(0, 1, num_steps, dtype=self.t_precision)
        lo, hi = min_t ** (1 / rho), max_t ** (1 / rho)
        self._sigmas = torch.flip((hi + ramp * (lo - hi)) ** rho, [0])
        self._min_step = int(min_step_percent * num_steps)
        self._max_step = int(max_step_percent * num_steps)

    @property
    def max_sigma(self) -> float:
        return self._max_t

    def alpha(self, t):
        return torch.ones_like(t)

    def get_t_list(self, sample_steps: int, device: Optional[torch.device] = None) -> torch.Tensor:
        """sample_steps+1 decreasing timesteps: table entries at linspace(max_step, min_step).long(), last := 0
        (noise_schedule.py:940-973)."""
        idx = torch.linspace(self._max_step, self._min_step, sample_steps + 1).long()
        t = self._sigmas[idx].clone()
        t[-1] = 0.0
        return t.to(device=device or self._sigmas.device, dtype=self.t_precision).clamp(max=self.max_t)

    def sample_t(self, n: int, time_dist_type: str = "polynomial", device=None, **kwargs) -> torch.Tensor:
        """Training-time timestep draws; only the table-index ('polynomial') and uniform forms are provided here."""
        if time_dist_type == "polynomial":
            idx = torch.randint(self._min_step, self._max_step + 1, (n,))
            t = self._sigmas[idx]
        elif time_dist_type == "uniform":
            t = torch.rand(n, dtype=self.t_precision) * (self.max_t - self.min_t) + self.min_t
        else:
            raise ValueError(f"Unsupported time distribution type: {time_dist_type} in EDMNoiseSchedule.")
        return t.to(device=device, dtype=self.t_precision).clamp(self.min_t, self.max_t)


class RFNoiseSchedule(BaseNoiseSchedule):
    """Rectified flow: x_t = (1 - t) x_0 + t eps, t = 0 is data, t in [0, 0.999] (noise_schedule.py:1306-1486)."""

    schedule_id = 1

    def __init__(self, min_t: float = 0.0, max_t: float = 0.999, num_steps: int = 1000, **kwargs):
        super().__init__(min_t, max_t, num_steps, **kwargs)
        assert 0 <= min_t < max_t <= 0.999, "RF min_t and max_t must be between 0 and 0.999"
        self._sigmas = torch.linspace(min_t, max_t, num_steps, dtype=self.t_precision)

    @property
    def max_sigma(self) -> float:
        return self._sigmas[int(self.num_steps * self.max_t)].item()

    def alpha(self, t):
        return 1 - t

    def sample_t(self, n: int, time_dist_type: str = "logitnormal", train_p_mean: float = 0, train_p_std: float = 1.0,
                 min_t: Optional[float] = 0.001, max_t: Optional[float] = 0.999, device=None, **kwargs) -> torch.Tensor:
        """Training-time timestep draws (noise_schedule.py:1383-1424)."""
        min_t = max(min_t, self.min_t) if min_t is not None else self.min_t
        max_t = min(max_t, self.max_t) if max_t is not None else self.max_t
        if time_dist_type == "logitnormal":
            t = torch.sigmoid(torch.randn(n, dtype=self.t_precision) * train_p_std + train_p_mean) * (max_t - min_t) + min_t
        elif time_dist_type in ("uniform", "shifted"):
            t = torch.rand(n, dtype=self.t_precision) * (max_t - min_t) + min_t
            if time_dist_type == "shifted":
                shift = kwargs.get("shift", 5.0)
                assert shift >= 1, f"shift must be >= 1, got {shift}"
                t = t * shift / (t * (shift - 1) + 1)
        else:
            raise ValueError(f"Unsupported time distribution type: {time_dist_type} in RFNoiseSchedule.")
        return t.to(device=device).clamp(min_t, max_t)


NOISE_SCHEDULES = {"edm": EDMNoiseSchedule, "rf": RFNoiseSchedule}


def get_noise_schedule(name: str, **kwargs):
    if name not in NOISE_SCHEDULES:
        raise KeyError(f"Unknown noise schedule '{name}'. Available schedules: {', '.join(sorted(NOISE_SCHEDULES))}")
    return NOISE_SCHEDULES[name](**kwargs)
