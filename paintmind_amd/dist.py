"""Multi-GPU generation: independent images, one process per GPU, no collective in the data path.

The reference has no multi-GPU inference; this is the new sharded path that
BASELINE.json asks for.  A global prompt list is split contiguously over ranks, every rank decodes its
own images with the sampling RNG keyed by the GLOBAL image index (so the result does not depend on the
number of ranks), and the finished images are gathered once with torch.distributed (RCCL on ROCm,
gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """contiguous, balanced split: the first (n % world) ranks get one extra item"""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_images(local, counts, dst=0, group=None):
    """Gather per-rank image tensors [n_r, ...] (ragged in dim 0) onto rank `dst`; returns the concatenation
    there and None elsewhere.  One collective per call; tensors are padded to the largest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nmax = max(counts)
    if local.is_cuda and dist.get_backend(group) != "nccl":
        local = local.cpu()                     # gloo groups (CPU tests, ranks sharing one GPU) exchange host tensors
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def gather_steps_to_host(local, counts, dst=0, group=None):
    """`local` [n_r, n_saved, ...] per rank -> on rank `dst` a HOST tensor [n_saved, sum(counts), ...], None elsewhere.
    One collective, like gather_images, but rank `dst` never holds more than ONE device copy of the whole result: the shards
    arrive in one preallocated [world, n_max, ...] buffer and each is copied straight into its rows of the (pinned, when the
    source is a GPU tensor) host result, already in step-major order.  (The concatenate-then-transpose form held the gather
    buffers, the concatenation and the transposed copy at once: three times the result, 40 GB for 8 ranks x 64 images x 9
    saved steps at 512 px.)"""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nmax = max(counts)
    if local.is_cuda and dist.get_backend(group) != "nccl":
        local = local.cpu()
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    big = torch.empty((world, nmax) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) if rank == dst else None
    dist.gather(pad.contiguous(), None if big is None else [big[r] for r in range(world)], dst=dst, group=group)
    del pad
    if rank != dst:
        return None
    n_saved = local.shape[1]
    out = torch.empty((n_saved, sum(counts)) + tuple(local.shape[2:]), dtype=local.dtype, pin_memory=local.is_cuda)
    off = 0
    for r, c in enumerate(counts):
        if c:
            out[:, off:off + c].copy_(big[r, :c].transpose(0, 1))     # one shard at a time: [c, n_saved, ...] -> rows of [n_saved, n, ...]
        off += c
    return out


def generate_sharded(pipe, text, seed, group=None, dst=0, timesteps=18, save_interval=2, **kwargs):
    """Pipeline.generate over a prompt list sharded across the process group.

    Returns on rank `dst` the same list-of-tensors structure the single-process call returns for the full
    prompt list (bit-identical for the same seed); other ranks get None.

    ONE collective per call: the images of all saved steps of a rank are stacked into one
    [n_local, n_saved, C, H, W] tensor and gathered once; rank `dst` copies shard after shard into the host result.  The
    number of saved steps depends only on (timesteps, save_interval) (generate.py:195-196) and the image shape on the
    pipeline (`pipe.image_shape`), so a rank whose shard is empty (fewer prompts than ranks) joins the same gather with
    a zero-row tensor without asking anybody.  Only a pipeline object that does not expose `image_shape` needs one small
    all_gather of the shape in that case -- which every rank then joins: rank-local facts never decide how many
    collectives a rank issues."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(len(text), r, world)[1] - shard_range(len(text), r, world)[0] for r in range(world)]
    lo, hi = shard_range(len(text), rank, world)
    n_out = sum(1 for step in range(timesteps) if step % save_interval == 0)
    if len(text) == 0:
        return [] if rank == dst else None
    tm = getattr(pipe, "text_model", None)
    if tm is not None and hasattr(tm, "base_index"):
        tm.base_index = lo                                   # synthetic text features are keyed by global index
    on = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    stacked = None
    if hi > lo:
        imgs = pipe.generate(text[lo:hi], timesteps=timesteps, save_interval=save_interval, seed=seed, image_base=lo,
                             keep_on_device=True, **kwargs)
        if len(imgs) != n_out:
            raise RuntimeError(f"generate returned {len(imgs)} images, the schedule implies {n_out}")
        stacked = torch.stack(list(imgs), dim=1)             # [n_local, n_saved, C, H, W]
    tail = getattr(pipe, "image_shape", None)
    if min(counts) == 0 and tail is None:
        mine = list(stacked.shape[2:]) if stacked is not None else []
        info = torch.tensor([len(mine)] + mine + [0] * (7 - len(mine)), dtype=torch.int64, device=on)
        allinfo = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info, group=group)
        src = next(a for a, c in zip(allinfo, counts) if c > 0).cpu().tolist()
        tail = src[1:1 + src[0]]
    if stacked is None:
        stacked = torch.zeros([0, n_out] + [int(d) for d in tail], dtype=torch.float32, device=on)
    g = gather_steps_to_host(stacked, counts, dst=dst, group=group)      # host [n_saved, n, C, H, W]
    del stacked
    if rank != dst:
        return None
    return list(g)                                           # contiguous (n, C, H, W) tensors, like the single-process call
