"""Multi-GPU sharding of the hot path: one process per GPU, units (cell, slot, transport block) are
independent, so they are dealt block-cyclically to ranks and NO collective sits in the data path. RCCL (the "nccl" backend
of torch.distributed on ROCm) is only used for the optional ingest scatter (codeword LLR slabs from one GPU to its peers over
the xGMI full mesh -- a root scatter drives all 7 links at once, no ring) and for the gather of the small result records
(CRC verdicts, iteration counts, decoded TB bytes). On CPU test runs the same code goes over gloo."""
import numpy as np


def assign(nof_units, world_size, rank, block=1):
    """Indices of the units rank `rank` owns: block-cyclic with `block` consecutive units per turn (a block keeps the
    codeblocks of one slot together so that the HARQ soft buffers of a (rnti, harq) stay on one GPU)."""
    idx = np.arange(nof_units)
    return idx[(idx // block) % world_size == rank]


def owner(unit, world_size, block=1):
    return (unit // block) % world_size


def scatter_units(payload, nof_units, src, group=None):
    """Ingest scatter: `payload` (on `src`: tensor [nof_units, ...]) -> this rank's units, in `assign` order."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = assign(nof_units, world, rank)
    shape = None
    if rank == src:
        shape = torch.tensor(list(payload.shape[1:]), dtype=torch.int64, device=payload.device)
        ndim = torch.tensor([shape.numel()], dtype=torch.int64, device=payload.device)
    dev = payload.device if payload is not None else None
    ndim_t = ndim if rank == src else torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(ndim_t, src, group=group)
    shape_t = shape if rank == src else torch.zeros(int(ndim_t.item()), dtype=torch.int64, device=dev)
    dist.broadcast(shape_t, src, group=group)
    tail = tuple(int(x) for x in shape_t.tolist())
    out = torch.empty((len(mine),) + tail, dtype=payload.dtype if rank == src else None, device=dev) if rank == src else None
    # Point-to-point sends (each peer has its own xGMI link to the source): no ring, no staging through third GPUs.
    if rank == src:
        reqs = []
        for r in range(world):
            sel = torch.as_tensor(assign(nof_units, world, r), device=payload.device)
            chunk = payload.index_select(0, sel).contiguous()
            if r == src:
                out = chunk
            else:
                reqs.append(dist.isend(chunk, r, group=group))
        for q in reqs:
            q.wait()
        return out
    raise RuntimeError("non-source ranks call recv_units()")


def recv_units(nof_units, src, dtype, device, group=None):
    """Counterpart of scatter_units on the receiving ranks."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = assign(nof_units, world, rank)
    ndim_t = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(ndim_t, src, group=group)
    shape_t = torch.zeros(int(ndim_t.item()), dtype=torch.int64, device=device)
    dist.broadcast(shape_t, src, group=group)
    out = torch.empty((len(mine),) + tuple(int(x) for x in shape_t.tolist()), dtype=dtype, device=device)
    dist.recv(out, src, group=group)
    return out


def gather_results(local, nof_units, group=None):
    """All-gather of per-unit result rows (local: [n_local, ...] in `assign` order) back into unit order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [len(assign(nof_units, world, r)) for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.empty((nof_units,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        sel = torch.as_tensor(assign(nof_units, world, r), device=local.device)
        out[sel] = bufs[r][:counts[r]]
    return out
