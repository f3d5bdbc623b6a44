"""Multi-GPU sharding of the hot path: one process per GPU, units (cell, slot, transport block) are
independent, so they are dealt block-cyclically to ranks and NO collective sits in the data path. RCCL (the "nccl" backend
of torch.distributed on ROCm) is only used for the optional ingest scatter (codeword LLR slabs from one GPU to its peers over
the xGMI full mesh -- the root posts one send per peer inside one group, so all 7 links carry data at once, no ring) and for
the gather of the small result records (CRC verdicts, iteration counts, decoded TB bytes). The reference's analogue is one
processor instance per worker (lib/phy/upper/uplink_processor_concurrent.h:41-54); here a worker is a GPU.

Every function is called by EVERY rank of the group with the same (nof_units, block) arguments. On a gloo group (CPU tests,
one-GPU rehearsal) device tensors are staged through host memory, since gloo's point-to-point path only takes host tensors."""
import numpy as np


def assign(nof_units, world_size, rank, block=1):
    """Indices of the units rank `rank` owns: block-cyclic with `block` consecutive units per turn (a block keeps the
    codeblocks of one slot together so that the HARQ soft buffers of a (rnti, harq) stay on one GPU)."""
    idx = np.arange(nof_units)
    return idx[(idx // block) % world_size == rank]


def owner(unit, world_size, block=1):
    return (unit // block) % world_size


def _host_staged(group):
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def scatter_units(payload, nof_units, src, unit_shape, dtype, device, block=1, group=None):
    """Ingest scatter. `payload` (only read on rank `src`): tensor [nof_units, *unit_shape]; every rank gets back its own
    units [len(assign(...)), *unit_shape] on `device`, in `assign` order. One grouped batch of point-to-point transfers:
    on RCCL that is ncclGroupStart / ncclSend x (world-1) / ncclGroupEnd on the source, a single ncclRecv elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = _host_staged(group)
    xdev = torch.device("cpu") if staged else device
    mine = assign(nof_units, world, rank, block)
    out = torch.empty((len(mine),) + tuple(unit_shape), dtype=dtype, device=xdev)
    ops, keep = [], []
    if rank == src:
        assert tuple(payload.shape) == (nof_units,) + tuple(unit_shape) and payload.dtype == dtype
        for r in range(world):
            sel = torch.as_tensor(assign(nof_units, world, r, block), device=payload.device)
            chunk = payload.index_select(0, sel)
            if r == src:
                out = chunk.to(device)
            elif chunk.shape[0]:
                chunk = chunk.to(xdev).contiguous()
                keep.append(chunk)
                ops.append(dist.P2POp(dist.isend, chunk, r, group))
    elif len(mine):
        ops.append(dist.P2POp(dist.irecv, out, src, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out.to(device)


def gather_results(local, nof_units, block=1, group=None):
    """All-gather of per-unit result rows (local: [n_local, ...] in `assign` order) back into unit order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    staged = _host_staged(group)
    counts = [len(assign(nof_units, world, r, block)) for r in range(world)]
    mx = max(counts)
    src = local.cpu() if staged else local
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=src.device)
    pad[:local.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.empty((nof_units,) + tuple(local.shape[1:]), dtype=local.dtype, device=src.device)
    for r in range(world):
        sel = torch.as_tensor(assign(nof_units, world, r, block), device=src.device)
        out[sel] = bufs[r][:counts[r]]
    return out.to(local.device)
