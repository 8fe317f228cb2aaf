"""`FastGenNetwork` duck type: the nn.Module surface the reference's methods / trainer / inference scripts
call (fastgen/networks/network.py:13-208).  The 'edm' and 'rf' schedules exist on this path."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Optional

import torch

from fastgen_amd.networks.noise_schedule import NET_PRED_TYPES, get_noise_schedule


class FastGenNetwork(ABC, torch.nn.Module):
    def __init__(self, net_pred_type: str = "x0", schedule_type: str = "edm", **net_kwargs):
        super().__init__()
        if net_pred_type not in NET_PRED_TYPES:
            raise ValueError(f"Unsupported net_pred_type '{net_pred_type}'. Supported types are: {NET_PRED_TYPES}")
        self.net_pred_type = net_pred_type
        self.schedule_type = schedule_type
        self.set_noise_schedule(**net_kwargs)

    def set_noise_schedule(self, schedule_type: Optional[str] = None, **kw) -> None:
        if schedule_type is not None:
            self.schedule_type = schedule_type
        self.noise_scheduler = get_noise_schedule(self.schedule_type, **kw)

    def reset_parameters(self):
        if getattr(self, "noise_scheduler", None) is not None:
            self.set_noise_schedule()

    def fully_shard(self, **kwargs):
        raise NotImplementedError(f"Network {self.__class__.__name__} does not implement the fully_shard method.")

    def sample(self, noise: torch.Tensor, condition: Optional[Any] = None, neg_condition: Optional[Any] = None,
               guidance_scale: Optional[float] = 5.0, num_steps: int = 50, **kwargs) -> torch.Tensor:
        raise NotImplementedError(f"Network {self.__class__.__name__} does not implement the sample method.")

    @abstractmethod
    def forward(self, x_t, t, condition=None, r=None, return_features_early=False, feature_indices=None,
                return_logvar=False, fwd_pred_type=None, **fwd_kwargs):
        ...
