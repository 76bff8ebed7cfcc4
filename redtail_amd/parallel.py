"""Multi-GPU plumbing of the Stereo DNN path: one process per GPU, stereo pairs sharded over the ranks,
and exactly one collective -- the broadcast of the weight-file image from rank 0 (RCCL over xGMI with
backend "nccl"; gloo in the CPU tests).  There is no data-path collective: pairs are independent."""
import torch


def broadcast_blob(blob, rank, device, dist):
    """Rank 0 passes the weight-file image (bytes); every rank returns the same bytes."""
    n = torch.tensor([len(blob) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, 0)
    return bytes(buf.cpu().numpy())


def shard(n_items, rank, world):
    """Contiguous block of item indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))
