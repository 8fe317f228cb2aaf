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

from fastgen_amd.methods.model import FastGenModel
from fastgen_amd.networks.noise_schedule import expand_like


class MeanFlowModel(FastGenModel):
    _fused_loop = "meanflow"

    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None,
                             student_sample_type: str = "sde", **kwargs) -> torch.Tensor:
        """x <- x - dt * u(x, t, r): 'sde' jumps to r = 0 with dt = t_cur and re-noises to t_next, 'ode' integrates the
        average velocity over [t_next, t_cur] (mean_flow.py:336-381)."""
        batch_size = x.shape[0]
        for t_cur, t_next in zip(t_list[:-1], t_list[1:]):
            t_cur_batch = t_cur.expand(batch_size)
            t_next_batch = t_next.expand(batch_size)
            if student_sample_type == "sde":
                delta_t = expand_like(t_cur, x).to(x.dtype)
                x = x - delta_t * net(x, t=t_cur_batch, condition=condition, r=torch.zeros_like(t_next_batch),
                                      fwd_pred_type="flow")
                if t_next > 0:
                    eps_infer = torch.randn_like(x)
                    x = net.noise_scheduler.forward_process(x, eps_infer, t_next_batch)
            elif student_sample_type == "ode":
                delta_t = expand_like(t_cur - t_next, x).to(x.dtype)
                x = x - delta_t * net(x, t=t_cur_batch, condition=condition, r=t_next_batch, fwd_pred_type="flow")
            else:
                raise NotImplementedError(
                    f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
        return x
