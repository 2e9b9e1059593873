"""Multi-GPU use of the operator: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The op is independent per (batch, q-head); smooth-K's mean is per (batch, kv-head) and every scale is per
(batch, head, block), so the work shards with NO data-path collective:
  * batch split (dim 0) into `world` contiguous chunks - contiguous views for both HND and NHD;
  * if B < world, kv-head groups are split instead (each rank takes Hkv/world kv heads and their q heads).
The only collective is the optional gather of the per-shard outputs (`all_gather_batch`), for callers that
need the whole tensor on every rank; data-parallel callers keep their shard and never communicate.
xGMI is a point-to-point mesh: a direct all-gather puts one peer's shard on each link, so the overlapped form
(`overlap=True`) gathers per batch element behind the next element's launch, on a cyclic batch partition whose
gathered pieces land in unsharded order.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def partition(n: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of `rank`'s contiguous share of n units; remainders go to the lowest ranks."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_spec(B: int, Hq: int, Hkv: int, world: int) -> str:
    """'batch' when there are at least `world` batch elements, else 'head' (kv-head groups), else error."""
    if B >= world:
        return "batch"
    if Hkv >= world:
        return "head"
    raise ValueError(f"cannot shard B={B}, Hkv={Hkv} over {world} ranks")


def shard_inputs(q, k, v, tensor_layout: str, world: int, rank: int):
    """Views (no copies) of this rank's share of q, k, v and the spec used."""
    hdim = 1 if tensor_layout == "HND" else 2
    B, Hq, Hkv = q.shape[0], q.shape[hdim], k.shape[hdim]
    spec = shard_spec(B, Hq, Hkv, world)
    if spec == "batch":
        s, e = partition(B, world, rank)
        return q[s:e], k[s:e], v[s:e], spec
    g = Hq // Hkv
    s, e = partition(Hkv, world, rank)
    sl_kv = [slice(None)] * 4
    sl_q = [slice(None)] * 4
    sl_kv[hdim] = slice(s, e)
    sl_q[hdim] = slice(s * g, e * g)
    return q[tuple(sl_q)], k[tuple(sl_kv)], v[tuple(sl_kv)], spec


def all_gather_batch(o, dim: int = 0, group=None):
    """All-gather equally- or unequally-sized shards along `dim` (RCCL all-gather when the backend is nccl)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return o
    sizes = [torch.zeros(1, dtype=torch.int64, device=o.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([o.shape[dim]], dtype=torch.int64, device=o.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    o = o.contiguous() if dim == 0 else o.transpose(0, dim).contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * sizes[0],) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
        dist.all_gather_into_tensor(out, o, group=group)
    else:  # uneven split: collectives want equal counts -> pad every shard to the largest, trim after
        nmax = max(sizes)
        padded = torch.zeros((nmax,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
        padded[: o.shape[0]] = o
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = torch.cat([part[:n] for part, n in zip(parts, sizes)], dim=0)
    return out if dim == 0 else out.transpose(0, dim)


def _pipelined_batch_gather(fn, q, k, v, tensor_layout, world, rank, group, kwargs):
    """Batch-sharded run with the gather issued behind the compute.  CYCLIC partition: rank r takes the batch elements
    r, r + world, r + 2 world, ... - strided views of q, k, v (the C ABI takes the batch stride as given, no copies) - and
    processes them one at a time; element i of every rank goes through ONE `all_gather_into_tensor` into the contiguous rows
    [i world, (i + 1) world) of the result while the next element's kernels run (RCCL works on its own stream).  Row
    i world + r is batch element i world + r: the staging buffer IS the result in unsharded order - no transposing copy, no
    second full-size buffer (rounds 1-3 split the batch into contiguous chunks and re-ordered the gathered pieces with a copy
    of the whole output, 8.6 GB read + written per rank at C5).  Needs B % world == 0.
    UNMEASURED on hardware so far (no multi-GPU node has been available): that only the last element's gather is exposed - on
    the xGMI mesh a direct all-gather moves one peer's piece per link - is an expectation, to be timed with `bench.py --gpus N`."""
    import torch
    import torch.distributed as dist
    B = q.shape[0]
    nb = B // world
    qs, ks, vs = q[rank::world], k[rank::world], v[rank::world]
    out = None
    handles = []
    for i in range(nb):
        o_i = fn(qs[i:i + 1], ks[i:i + 1], vs[i:i + 1], tensor_layout=tensor_layout, **kwargs)
        if out is None:
            out = torch.empty((B,) + tuple(o_i.shape[1:]), dtype=o_i.dtype, device=o_i.device)
        handles.append(dist.all_gather_into_tensor(out[i * world:(i + 1) * world], o_i.contiguous(), group=group, async_op=True))
    for h in handles:
        h.wait()
    return out


def sharded_attention(fn: Callable, q, k, v, *, tensor_layout: str = "HND", gather: bool = True, group=None,
                      overlap: bool = False, **kwargs):
    """Run `fn` (one of the lowbit_fa_* operators) on this rank's shard of replicated q, k, v.
    gather=True: return the full output on every rank (one all-gather); False: return the local shard.
    overlap=True (batch sharding with equal shards, no lse): gather piecewise behind the compute."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return fn(q, k, v, tensor_layout=tensor_layout, **kwargs)
    qs, ks, vs, spec = shard_inputs(q, k, v, tensor_layout, world, rank)
    if gather and overlap and spec == "batch" and q.shape[0] % world == 0 and not kwargs.get("return_lse", False):
        return _pipelined_batch_gather(fn, q, k, v, tensor_layout, world, rank, group, kwargs)
    out = fn(qs, ks, vs, tensor_layout=tensor_layout, **kwargs)
    if not gather:
        return out
    hdim = 1 if tensor_layout == "HND" else 2
    gdim = 0 if spec == "batch" else hdim
    if isinstance(out, tuple):  # (o, lse): lse is [B, Hq, S]
        o, lse = out
        return all_gather_batch(o, gdim, group), all_gather_batch(lse, 0 if spec == "batch" else 1, group)
    return all_gather_batch(out, gdim, group)
