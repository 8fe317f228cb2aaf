"""`MeanFlowModel` sampling entry points with the reference's signatures
(fastgen/methods/consistency_model/mean_flow.py:51, 336-381; `generator_fn` is inherited from FastGenModel,
methods/model.py:374-420).

For a flow-predicting `r_timestep` `fastgen_amd` EDM network (configs/experiments/EDM/config_mf_cifar10.py) the whole
loop is one `fg_sampler_run(..., FG_LOOP_MEANFLOW)` call; anything else takes the per-step loop below, which is the
reference's.
"""
from __future__ import annotations

from typing import Any

import torch

from fastgen_amd.methods.model import _FUSED_LOOPS, FastGenModel
from fastgen_amd.networks.noise_schedule import expand_like


class MeanFlowModel(FastGenModel):
    _fused_loop = "meanflow"

    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None,
                             student_sample_type: str = "sde", **kwargs) -> torch.Tensor:
        """x <- x - dt * u(x, t, r) per interval of t_list (mean_flow.py:336-381).
        'sde': r = 0, dt = t (a jump to the data end), then re-noise to the next timestep unless it is 0;
        'ode': r = next timestep, dt = t - r (the average velocity integrates the interval exactly)."""
        if student_sample_type not in ("sde", "ode"):
            raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
        n = x.shape[0]
        jump = student_sample_type == "sde"
        for i in range(len(t_list) - 1):
            t, t_to = t_list[i], t_list[i + 1]
            r = torch.zeros_like(t_to) if jump else t_to
            dt = expand_like(t if jump else t - t_to, x).to(x.dtype)
            x = x - dt * net(x, t=t.expand(n), condition=condition, r=r.expand(n), fwd_pred_type="flow")
            if jump and t_to > 0:
                x = net.noise_scheduler.forward_process(x, torch.randn_like(x), t_to.expand(n))
        return x

    @staticmethod
    def network_jvp(net, x_t: torch.Tensor, t: torch.Tensor, r: torch.Tensor, dxt_dt: torch.Tensor,
                    condition: Any = None) -> torch.Tensor:
        """The tangent term of the MeanFlow objective, d/dt u(x_t + s dxt_dt, t + s, r) at s = 0 - the JVP branch of the
        reference's `MeanFlowModel._jvp` (mean_flow.py:240-250: `torch.func.jvp(net_wrapper, (x_t, t, r), (dxt_dt, 1, 0))`), as
        one `fg_edm_jvp` call for a `fastgen_amd` network.  Detached, like the reference's use of it.  It shares the module's
        training workspace: call it BEFORE the differentiable forward of the same step (otherwise that forward's kept state is
        overwritten and the backward recomputes it - correct, but one forward slower)."""
        if not hasattr(net, "jvp"):
            raise NotImplementedError("network_jvp needs a fastgen_amd network (EDMPrecond.jvp)")
        _, u_jvp = net.jvp(x_t, t, dxt_dt, torch.ones_like(t, dtype=torch.float32), condition=condition, r=r,
                           v_r=torch.zeros_like(r, dtype=torch.float32), fwd_pred_type="flow" if net.net_pred_type == "flow" else None)
        return u_jvp


_FUSED_LOOPS.add(MeanFlowModel._student_sample_loop.__func__)
