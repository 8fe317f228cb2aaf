"""`TrigFlowPrecond` for a `fastgen_amd` denoiser, with the reference's interface (fastgen/methods/consistency_model/sCM.py:21-83):
the sCM-family trainers (sCM / TCM / sCD) talk to the network through this wrapper - TrigFlow time t_hat in (0, pi/2), inputs
scaled to the denoiser's schedule by SNR matching, output F_theta = flow(x_t_hat, x0_pred, t_hat) / sigma_data - and take its
forward-mode derivative (`torch.func.jvp`, sCM.py:150-181).

Everything around the network call is elementwise torch on [B]-sized or image-sized tensors (host-side plumbing, as in the
reference); the network itself is the HIP engine.  `jvp()` composes the wrapper's derivative from three pieces: `torch.func.jvp`
of the input map, `EDMPrecond.jvp` (one fg_edm_jvp call) and `torch.func.jvp` of the output map - a custom-kernel module cannot be
traced by `torch.func`, so `SCMModel._jvp`'s single call site is redirected here."""
from __future__ import annotations

from typing import Any, Optional

import torch

from fastgen_amd.networks.noise_schedule import expand_like, get_noise_schedule


class TrigFlowPrecond(torch.nn.Module):
    def __init__(self, net, sigma_data: float = 0.5):
        super().__init__()
        self.net = net
        self.sigma_data = sigma_data
        self.net_pred_type, self.schedule_type = "flow", "trig"
        self.noise_scheduler = get_noise_schedule("trig")

    def _convert_trigflow_to_net_input(self, x_t_hat: torch.Tensor, t_hat: torch.Tensor):
        """(x_t, t) of the denoiser's schedule with the same signal-to-noise ratio as (x_t_hat, t_hat) in TrigFlow, fp64 inside
        (sCM.py:35-58)."""
        x64, t64 = x_t_hat.double(), t_hat.double()
        t = self.net.noise_scheduler.sqrt_snr_to_t(self.noise_scheduler.sqrt_snr(t64) / self.sigma_data)
        a, s = self.net.noise_scheduler.alpha(t), self.net.noise_scheduler.sigma(t)
        coeff = (a**2 + (s / self.sigma_data) ** 2).sqrt()
        return (x64 * expand_like(coeff, x64)).to(x_t_hat.dtype), t.to(t_hat.dtype)

    def _output(self, x_t_hat, x0_pred, t_hat):
        return self.noise_scheduler.x0_to_flow(x_t_hat, x0_pred, t_hat) / self.sigma_data

    def forward(self, x_t_hat, t_hat, condition: Any = None, return_logvar: bool = False, return_x0_pred: bool = False, **kw):
        x_t, t = self._convert_trigflow_to_net_input(x_t_hat, t_hat)
        outs = self.net(x_t, t, condition=condition, return_logvar=return_logvar, fwd_pred_type="x0", **kw)
        x0_pred, logvar = (outs[0], outs[1]) if return_logvar else (outs, None)
        F_theta = self._output(x_t_hat, x0_pred, t_hat)
        if return_x0_pred and return_logvar:
            return F_theta, logvar, x0_pred
        if return_x0_pred:
            return F_theta, x0_pred
        if return_logvar:
            return F_theta, logvar
        return F_theta

    @torch.no_grad()
    def jvp(self, x_t_hat: torch.Tensor, t_hat: torch.Tensor, v_x: torch.Tensor, v_t: torch.Tensor, condition: Any = None,
            eps: Optional[float] = 1e-4):
        """(F_theta, its directional derivative along (v_x, v_t)) - `torch.func.jvp(net_trigflow_wrapper, (x_t_hat, t_hat),
        (v_x, v_t))` of SCMModel._jvp, including its clamp of t_hat away from +-pi/2 (sCM.py:160-179)."""
        def pre(xh, th):
            if eps is not None:
                th = th.clamp(min=-torch.pi / 2 + eps, max=torch.pi / 2 - eps)
            return self._convert_trigflow_to_net_input(xh, th)

        def post(xh, x0, th):
            if eps is not None:
                th = th.clamp(min=-torch.pi / 2 + eps, max=torch.pi / 2 - eps)
            return self._output(xh, x0, th)

        v_t = v_t.to(t_hat.dtype)
        (x_t, t), (dx_t, dt) = torch.func.jvp(pre, (x_t_hat, t_hat), (v_x, v_t))
        x0, dx0 = self.net.jvp(x_t, t, dx_t, dt, condition=condition, fwd_pred_type="x0")
        return torch.func.jvp(post, (x_t_hat, x0, t_hat), (v_x, dx0.to(x0.dtype), v_t))
