"""Host-side noise schedules exposed as `net.noise_scheduler`: EDM ('edm') and rectified flow ('rf').

Mirrors the surface of the reference's `BaseNoiseSchedule` / `EDMNoiseSchedule` / `RFNoiseSchedule`
(fastgen/networks/noise_schedule.py:23-726, 729-1035, 1306-1486) that the sampling callers read
(methods/model.py:361-413, methods/consistency_model/mean_flow.py:336-381): get_t_list, latents, forward_process,
x0_to_eps, convert_model_output, max_t, max_sigma, t_precision, sigmas, is_t_valid, sample_t — and the ones the training
callers read (methods/distribution_matching/dmd2.py:100-120, consistency_model/*.py): sample_from_t_list,
next_in_t_list, rescale_t, alpha_prime / sigma_prime, cond_velocity, sqrt_snr(_to_t), closest_sigma_idx,
sigma_idx_to_t, safe_clamp.  These are tiny tensor
expressions evaluated with torch on whatever device the inputs live on; inside the fused sampler (fg_sampler_run)
the same formulas run as HIP kernels (csrc/misc.hip) and these classes only supply the timestep list.
"""
from __future__ import annotations

import math
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

    def alpha_prime(self, t):
        raise NotImplementedError

    def sigma_prime(self, t):
        """d sigma / dt = 1 for both schedules (noise_schedule.py:782-783, 1346-1347)."""
        return torch.ones_like(t)

    def rescale_t(self, t: torch.Tensor) -> torch.Tensor:
        """The timestep as the network consumes it (noise_schedule.py:140-148)."""
        assert self.is_t_valid(t), f"t must be in range [{self.min_t}, {self.max_t}], but got {t}"
        return self._rescale(t)

    # -- helpers ---------------------------------------------------------------------------------------
    def is_t_valid(self, t: torch.Tensor) -> torch.Tensor:
        """min_t <= t <= max_t up to one ulp of t's dtype (noise_schedule.py:409-423)."""
        dt = t.dtype if t.dtype in (torch.bfloat16, torch.float32, torch.float64) else torch.float32
        lo = torch.nextafter(torch.tensor(self.min_t, dtype=dt, device=t.device), torch.tensor(-float("inf"), dtype=dt, device=t.device))
        hi = torch.nextafter(torch.tensor(self.max_t, dtype=dt, device=t.device), torch.tensor(float("inf"), dtype=dt, device=t.device))
        return torch.all((lo <= t) & (t <= hi))

    def non_zero_clamp(self, x: torch.Tensor) -> torch.Tensor:
        return torch.where(x >= 0, x.clamp(min=self.clamp_min), x.clamp(max=-self.clamp_min))

    @staticmethod
    def safe_clamp(t: torch.Tensor, min: Optional[float] = None, max: Optional[float] = None) -> torch.Tensor:
        """clamp whose bounds are first moved inwards to values representable in t's dtype, so that
        min <= result <= max holds exactly (noise_schedule.py:90-121)."""
        def inward(bound, towards):
            b = torch.as_tensor(bound, dtype=t.dtype, device=t.device)
            crossed = b.item() < bound if towards > 0 else b.item() > bound
            if crossed:
                b = torch.nextafter(b, torch.tensor(towards * float("inf"), dtype=t.dtype, device=t.device))
            return b.item()

        lo = inward(min, +1) if min is not None else None
        hi = inward(max, -1) if max is not None else None
        return torch.clamp(t, min=lo, max=hi)

    # -- what the samplers call ------------------------------------------------------------------------------
    def get_t_list(self, sample_steps: int, device: Optional[torch.device] = None) -> torch.Tensor:
        """linspace(max_t, 0, sample_steps+1) (noise_schedule.py:259-272)."""
        t = torch.linspace(self.max_t, 0, sample_steps + 1, device=device or self._sigmas.device, dtype=self.t_precision)
        return t.clamp(max=self.max_t)

    def _resolve_t_list(self, sample_steps: int, t_list, device) -> torch.Tensor:
        if t_list is None:
            return self.get_t_list(sample_steps=sample_steps, device=device or self._sigmas.device)
        return torch.as_tensor(t_list, device=device, dtype=self.t_precision)

    def sample_from_t_list(self, n: int, sample_steps: int, t_list=None, return_ids: Optional[bool] = False,
                           device: Optional[torch.device] = None):
        """n uniform draws from t_list[:-1] — the final (clean, t = 0) entry is never trained on
        (noise_schedule.py:274-304).  The index draw uses the CPU generator, like the reference's."""
        tl = self._resolve_t_list(sample_steps, t_list, device)
        ids = torch.randint(0, len(tl) - 1, (n,)).to(device=device)
        return (tl[ids], ids) if return_ids else tl[ids]

    def next_in_t_list(self, ids: torch.Tensor, sample_steps: int, t_list, device: Optional[torch.device] = None,
                       stride: int = 1) -> torch.Tensor:
        """t_list[ids + stride]; running past the end is an error, not a clamp (noise_schedule.py:306-340)."""
        tl = self._resolve_t_list(sample_steps, t_list, device)
        if t_list is not None:
            assert tl.shape == (sample_steps + 1,), f"t_list must be of shape (sample_steps + 1,), but got {tl.shape}"
        nxt = ids + stride
        if nxt.max() > sample_steps:
            raise ValueError(f"Clamping next ids to sample steps. ids: {ids}, next_ids: {nxt}, sample_steps: {sample_steps}")
        return tl[nxt]

    def cond_velocity(self, x: torch.Tensor, eps: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """dx_t/dt = alpha'(t) x_0 + sigma'(t) eps in fp64, cast back (noise_schedule.py:451-476)."""
        assert self.is_t_valid(t), f"t must be in [{self.min_t}, {self.max_t}], but got {t}"
        t64 = t.to(torch.float64)
        out = x.to(torch.float64) * expand_like(self.alpha_prime(t64), x) + eps.to(torch.float64) * expand_like(
            self.sigma_prime(t64), eps)
        return out.to(x.dtype)

    def closest_sigma_idx(self, sigma_t: torch.Tensor) -> torch.Tensor:
        """Index of the table entry nearest to each sigma; ties go to the lower index (noise_schedule.py:478-505)."""
        shape = sigma_t.shape
        flat = sigma_t.reshape(shape[0]) if sigma_t.ndim > 1 else sigma_t
        table = self.sigmas.to(flat)
        hi = torch.searchsorted(table, flat, side="right")
        lo = (hi - 1).clamp(min=0)
        hi = hi.clamp(max=table.numel() - 1)
        pick_hi = (table[hi] - flat).abs() < (table[lo] - flat).abs()
        return torch.where(pick_hi, hi, lo).view(shape)

    def sigma_idx_to_t(self, sigma_idx: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def sqrt_snr(self, t: torch.Tensor) -> torch.Tensor:
        """alpha(t) / clamp(sigma(t)), evaluated and returned in fp64 (noise_schedule.py:518-531)."""
        assert self.is_t_valid(t)
        t64 = t.to(torch.float64)
        return self.alpha(t64) / self.non_zero_clamp(self.sigma(t64))

    def sqrt_snr_to_t(self, sqrt_snr_t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

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

    def alpha_prime(self, t):
        return torch.zeros_like(t)

    def _rescale(self, t):
        return t

    def sigma_idx_to_t(self, sigma_idx: torch.Tensor) -> torch.Tensor:
        """sigma(t) = t: the table entry itself (noise_schedule.py:785-799)."""
        assert sigma_idx.dtype == torch.long
        return self._sigmas.to(device=sigma_idx.device)[sigma_idx]

    def sqrt_snr_to_t(self, sqrt_snr_t: torch.Tensor) -> torch.Tensor:
        """alpha / sigma = 1 / t  =>  t = 1 / clamp(sqrt_snr), in fp64 (noise_schedule.py:801-817)."""
        return (1 / self.non_zero_clamp(sqrt_snr_t.to(torch.float64))).to(sqrt_snr_t.dtype)

    def get_t_list(self, sample_steps: int, device: Optional[torch.device] = None) -> torch.Tensor:
        """sample_steps+1 decreasing timesteps: table entries at linspace(max_step, min_step).long(), last := 0
        (noise_schedule.py:940-973)."""
        idx = torch.linspace(self._max_step, self._min_step, sample_steps + 1).long()
        t = self._sigmas[idx].clone()
        t[-1] = 0.0
        return t.to(device=device or self._sigmas.device, dtype=self.t_precision).clamp(max=self.max_t)

    def sample_t(self, n: int, time_dist_type: str = "polynomial", train_p_mean: float = -1.2, train_p_std: float = 1.2,
                 min_t: Optional[float] = 0.002, max_t: Optional[float] = 80.0, device=None, **kwargs) -> torch.Tensor:
        """Training-time timestep draws (noise_schedule.py:878-938): 'polynomial' (uniform over the table indices
        [min_step, max_step]), 'uniform', 'lognormal' (ln t ~ N(mean, std) truncated to [min_t, max_t], by inverse CDF in
        t_precision on the CPU generator, :819-843).  The Student-t form ('log_t', scipy) is not provided."""
        min_t = max(min_t, self.min_t) if min_t is not None else self.min_t
        max_t = min(max_t, self.max_t) if max_t is not None else self.max_t
        dev = device or self._sigmas.device
        if time_dist_type == "polynomial":
            idx = torch.randint(self._min_step, self._max_step + 1, (n,), device=self._sigmas.device)
            t = self._sigmas[idx].to(device=dev, dtype=self.t_precision)
        elif time_dist_type == "uniform":
            t = torch.rand(n, device=dev, dtype=self.t_precision) * (max_t - min_t) + min_t
        elif time_dist_type == "lognormal":
            normal = torch.distributions.Normal(torch.tensor(train_p_mean, dtype=self.t_precision),
                                                torch.tensor(train_p_std, dtype=self.t_precision))
            c_lo = normal.cdf(torch.tensor(max(min_t, self.clamp_min), dtype=self.t_precision).log())
            c_hi = normal.cdf(torch.tensor(max_t, dtype=self.t_precision).log())
            t = normal.icdf(torch.rand(n, dtype=self.t_precision) * (c_hi - c_lo) + c_lo).exp().to(device=dev)
        else:
            raise ValueError(f"Unsupported time distribution type: {time_dist_type} in EDMNoiseSchedule.")
        return self.safe_clamp(t, min_t, max_t)


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

    def alpha_prime(self, t):
        return -torch.ones_like(t)

    def _rescale(self, t):
        return self.num_steps * t

    def sigma_idx_to_t(self, sigma_idx: torch.Tensor) -> torch.Tensor:
        """index / num_steps (noise_schedule.py:1349-1362)."""
        assert sigma_idx.dtype == torch.long
        return sigma_idx.to(self.t_precision) / self.num_steps

    def sqrt_snr_to_t(self, sqrt_snr_t: torch.Tensor) -> torch.Tensor:
        """(1 - t) / t = s  =>  t = 1 / (s + 1), in fp64 (noise_schedule.py:1364-1381)."""
        return (1 / (sqrt_snr_t.to(torch.float64) + 1)).to(sqrt_snr_t.dtype)

    def sample_t(self, n: int, time_dist_type: str = "logitnormal", train_p_mean: float = 0, train_p_std: float = 1.0,
                 min_t: Optional[float] = 0.001, max_t: Optional[float] = 0.999, device=None, **kwargs) -> torch.Tensor:
        """Training-time timestep draws (noise_schedule.py:1383-1424)."""
        min_t = max(min_t, self.min_t) if min_t is not None else self.min_t
        max_t = min(max_t, self.max_t) if max_t is not None else self.max_t
        dev = device or self._sigmas.device
        if time_dist_type == "logitnormal":
            t = torch.sigmoid(torch.randn(n, device=dev, dtype=self.t_precision) * train_p_std + train_p_mean) * (max_t - min_t) + min_t
        elif time_dist_type in ("uniform", "shifted"):
            t = torch.rand(n, device=dev, dtype=self.t_precision) * (max_t - min_t) + min_t
            if time_dist_type == "shifted":
                shift = kwargs.get("shift", 5.0)
                assert shift >= 1, f"shift must be >= 1, got {shift}"
                t = t * shift / (t * (shift - 1) + 1)
        else:
            raise ValueError(f"Unsupported time distribution type: {time_dist_type} in RFNoiseSchedule.")
        return self.safe_clamp(t, min_t, max_t)


class TrigNoiseSchedule(BaseNoiseSchedule):
    """TrigFlow: x_t = cos(t) x_0 + sin(t) eps, t in [0, pi/2] (noise_schedule.py:1489-1648) - the time axis of the sCM-family
    trainers (consistency_model/sCM.py), which wrap an EDM denoiser in `TrigFlowPrecond`.  Host-side only: nothing here runs
    inside the fused sampler."""

    schedule_id = -1  # no fused loop on this schedule

    def __init__(self, min_t: float = 0.0, max_t: float = math.pi / 2, num_steps: int = 1000, **kwargs):
        super().__init__(min_t, max_t, num_steps, **kwargs)
        assert 0 <= min_t < max_t, "Trig min_t must be non-negative and less than max_t"
        self._sigmas = torch.sin(torch.linspace(min_t, max_t, num_steps, dtype=self.t_precision))

    @property
    def max_sigma(self) -> float:
        return torch.sin(torch.tensor(self.max_t)).item()

    def alpha(self, t):
        return torch.cos(t)

    def sigma(self, t):
        return torch.sin(t)

    def alpha_prime(self, t):
        return -torch.sin(t)

    def sigma_prime(self, t):
        return torch.cos(t)

    def _rescale(self, t):
        return t

    def sigma_idx_to_t(self, sigma_idx: torch.Tensor) -> torch.Tensor:
        """The table is sin of a linspace in t: index -> t linearly (noise_schedule.py:1532-1545)."""
        assert sigma_idx.dtype == torch.long
        return sigma_idx.to(self.t_precision) / (self.num_steps - 1) * (self.max_t - self.min_t) + self.min_t

    def sqrt_snr(self, t: torch.Tensor) -> torch.Tensor:
        """cot(t) with tan clamped away from zero, fp64 (noise_schedule.py:1547-1557)."""
        assert self.is_t_valid(t)
        return 1.0 / self.non_zero_clamp(torch.tan(t.to(torch.float64)))

    def sqrt_snr_to_t(self, sqrt_snr_t: torch.Tensor) -> torch.Tensor:
        """arccot as atan2(1, s), fp64 (noise_schedule.py:1559-1574)."""
        s64 = sqrt_snr_t.to(torch.float64)
        return torch.atan2(torch.ones_like(s64), s64).to(sqrt_snr_t.dtype)

    def sample_t(self, n: int, time_dist_type: str = "uniform", train_p_mean: float = 0, train_p_std: float = 1.0,
                 min_t: Optional[float] = 0.0, max_t: Optional[float] = math.pi / 2, device=None, **kwargs) -> torch.Tensor:
        """Training-time timestep draws (noise_schedule.py:1576-1612)."""
        min_t = max(min_t, self.min_t) if min_t is not None else self.min_t
        max_t = min(max_t, self.max_t) if max_t is not None else self.max_t
        dev = device or self._sigmas.device
        if time_dist_type == "logitnormal":
            t = torch.sigmoid(torch.randn(n, device=dev, dtype=self.t_precision) * train_p_std + train_p_mean) * (max_t - min_t) + min_t
        elif time_dist_type == "uniform":
            t = torch.rand(n, device=dev, dtype=self.t_precision) * (max_t - min_t) + min_t
        else:
            raise ValueError(f"Unsupported time distribution type: {time_dist_type} in TrigNoiseSchedule.")
        return self.safe_clamp(t, min_t, max_t)

    def flow_to_x0(self, xt, v, t):
        """x_0 = cos(t) x_t - sin(t) v, fp64 (noise_schedule.py:1614-1632)."""
        assert self.is_t_valid(t), f"t must be in [{self.min_t}, {self.max_t}], but got {t}"
        t64 = t.to(torch.float64)
        out = xt.to(torch.float64) * expand_like(torch.cos(t64), xt) - v.to(torch.float64) * expand_like(torch.sin(t64), xt)
        return out.to(xt.dtype)

    def x0_to_flow(self, xt, x0, t):
        """v = (cos(t) x_t - x_0) / clamp(sin(t)), fp64 (noise_schedule.py:1634-1648)."""
        assert self.is_t_valid(t), f"t must be in [{self.min_t}, {self.max_t}], but got {t}"
        t64 = t.to(torch.float64)
        num = xt.to(torch.float64) * expand_like(torch.cos(t64), xt) - x0.to(torch.float64)
        return (num / self.non_zero_clamp(expand_like(torch.sin(t64), xt))).to(xt.dtype)


NOISE_SCHEDULES = {"edm": EDMNoiseSchedule, "rf": RFNoiseSchedule, "rectified_flow": RFNoiseSchedule, "trig": TrigNoiseSchedule}


def get_noise_schedule(name: str, **kwargs):
    if name not in NOISE_SCHEDULES:
        raise KeyError(f"Unknown noise schedule '{name}'. Available schedules: {', '.join(sorted(NOISE_SCHEDULES))}")
    return NOISE_SCHEDULES[name](**kwargs)
