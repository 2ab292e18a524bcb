"""Multi-GPU data parallelism for the inference path: one process per GPU,
images sharded by contiguous batch slices, weights replicated, no
communication in forward / decode / NMS.  The only collective is the gather of
the variable-length (K_r, 9) detection rows (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all; its per-image loop
(/root/reference/utilis.py:417) is what makes images independent.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous slice [start, stop) of `total` images owned by `rank`; the
    first total % world ranks get one extra image."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_detections(output, image_offset, device=None, group=None):
    """All-gather per-rank detections into the single-process result.

    output: the (K, 9) tensor filter_results returned on this rank, or the int
    0 it returns when nothing passed.  image_offset: index of this rank's
    first image in the global batch; it is added to column 0 so the gathered
    rows are exactly what one process would have produced for the whole batch
    (ranks hold ascending contiguous slices, so concatenating in rank order
    keeps the image-ascending order).

    Two collectives: an all-gather of the row counts, then an all-gather of the
    rows padded to the largest count.  Messages are small (tens of KB to a few
    MB), so this is latency-bound; padded all-gather keeps it to two launches.
    Returns the gathered (sum K, 9) tensor on every rank, or 0 if all are empty.
    """
    world = dist.get_world_size(group)
    if isinstance(output, int):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) \
                if dist.get_backend(group) == "nccl" else torch.device("cpu")
        rows = torch.zeros((0, 9), dtype=torch.float32, device=device)
    else:
        rows = output.clone()
        rows[:, 0] += float(image_offset)
        device = rows.device
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = counts.tolist()
    kmax = max(counts)
    if kmax == 0:
        return 0
    padded = torch.zeros((kmax, 9), dtype=torch.float32, device=device)
    padded[:rows.shape[0]] = rows
    gathered = torch.empty((world * kmax, 9), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    pieces = [gathered[r * kmax:r * kmax + counts[r]] for r in range(world) if counts[r]]
    return torch.cat(pieces, 0)
