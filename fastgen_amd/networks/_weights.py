"""Parameter hand-over of the transformer networks (DiT, CausalWan) to their engines, plain or under FSDP2.

`fg_dit_pack_group` / `fg_wan_pack_group` COPY what they are given (block linears into GEMM layouts, everything else into engine-owned
fp32 storage), so a parameter only has to be whole while its group is packed.  That is what lets the reference's sharded data
parallelism (fastgen/utils/distributed/fsdp.py:152-180 calling `net.fully_shard`, e.g. DiT/network.py:402-420, Wan/network.py:761-782) work
without ever holding more than ONE group's all-gather: per weight version, group by group - all-gather (the next group's already in
flight), bind, pack, reshard.  On MI355X the packed compute copy (bf16: 2 bytes per parameter, 28 GB for a 14B network of 288 GB) stays
whole on every GPU; the fp32 master parameters, gradients and optimizer state are what FSDP2 shards.  A frozen network (teacher, sampling)
gathers each group exactly once.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch


def _fsdp_module_type():
    try:
        from torch.distributed.fsdp import FSDPModule

        return FSDPModule
    except ImportError:  # pragma: no cover
        return None


def _local(p: torch.Tensor) -> torch.Tensor:
    return getattr(p, "_local_tensor", p)


def weight_groups(owner: torch.nn.Module, names: List[str]) -> List[Tuple[str, Optional[str], Optional[torch.nn.Module], List[str]]]:
    """(prefix, exclude, fsdp_module | None, parameter names) - one entry per FSDP2 parameter group of the tree (a module wrapped by
    `fully_shard` owns the parameters below it that no nested wrapped module owns), then one for whatever is left (plain tensors)."""
    FSDPModule = _fsdp_module_type()
    wrapped = [(n, m) for n, m in owner.named_modules() if FSDPModule is not None and isinstance(m, FSDPModule)]
    if any(n == "" for n, _ in wrapped):
        raise NotImplementedError("fully_shard applied to the network object itself; the reference shards its blocks / submodules "
                                  "(use net.fully_shard(**kwargs))")
    groups, taken = [], set()
    for n, m in sorted(wrapped, key=lambda nm: -len(nm[0])):  # innermost first
        pre = n + "."
        nested = sorted({w + "." for w, _ in wrapped if w.startswith(pre)})
        if len(nested) > 1 and len({x.rsplit(".", 2)[0] for x in nested}) != 1:
            raise NotImplementedError(f"FSDP groups nested under {n!r} do not share one prefix: {nested[:3]} ...")
        exclude = None
        if nested:
            exclude = nested[0].rsplit(".", 2)[0] + "."  # e.g. "transformer.blocks."
        mine = [k for k in names if k.startswith(pre) and k not in taken and not (exclude and k.startswith(exclude))]
        taken.update(mine)
        groups.append((pre, exclude, m, mine))
    rest = [k for k in names if k not in taken]
    if wrapped:
        groups.extend((k, None, None, [k]) for k in rest)  # (a name is its own prefix)
    else:
        groups.append(("", None, None, rest))
    return [g for g in groups if g[3]]


def sync_weights(owner: torch.nn.Module, names: List[str], sigs: Dict[str, tuple], tensors: Callable[[], Dict[str, torch.Tensor]],
                 bind: Callable[[str, torch.Tensor], None], pack_group: Callable[[str, Optional[str]], None]) -> bool:
    """Bring the engine's copies up to date: every group whose parameters changed (storage, in-place version, dtype) since its last
    pack is gathered (if sharded), bound and packed.  Returns True if anything was packed."""
    groups = weight_groups(owner, names)
    cur = tensors()
    todo = []
    for pre, exc, mod, mine in groups:
        sig = tuple((_local(cur[n]).data_ptr(), cur[n]._version, cur[n].dtype, tuple(cur[n].shape)) for n in mine)
        if sigs.get(pre) != sig:
            todo.append((pre, exc, mod, mine, sig))
    if not todo:
        return False
    pending = None  # the all-gather of the next sharded group, issued before the current one is packed
    for i, (pre, exc, mod, mine, sig) in enumerate(todo):
        if mod is not None:
            # (FSDP2's all-gather buffers must be ordinary tensors: created under torch.inference_mode() they could not be reused - or
            # version-checked - by a later gather outside it)
            with torch.inference_mode(False), torch.no_grad():
                if pending is not None and pending[0] is mod:
                    if pending[1] is not None:
                        pending[1].wait()
                else:
                    mod.unshard()
                pending = None
                nxt = next((t[2] for t in todo[i + 1:] if t[2] is not None), None)
                if nxt is not None:
                    pending = (nxt, nxt.unshard(async_op=True))
            cur = tensors()  # the group's parameters are whole tensors now
        keep = []
        for n in mine:
            p = cur[n]
            if p.device.type != "cuda":
                raise RuntimeError(f"parameter {n} is on {p.device}; fastgen_amd runs on a HIP GPU only (no CPU path)")
            q = _local(p.detach())
            if q.dtype != torch.float32 or not q.is_contiguous():
                q = q.to(torch.float32).contiguous()
            keep.append(q)
            bind(n, q)
        pack_group(pre, exc)  # copies on the current stream: `keep` and the gathered parameters may go afterwards (stream-ordered frees)
        if mod is not None:
            with torch.inference_mode(False), torch.no_grad():
                mod.reshard()
        sigs[pre] = sig
    return True
