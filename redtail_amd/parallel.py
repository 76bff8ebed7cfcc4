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


def exchange_unique_id(lib, rank, world, dist=None):
    """The 128-byte RCCL unique id of a communicator of our own: rank 0 makes it (rt_comm_unique_id), the others receive it through `dist`
    (any initialised torch.distributed backend; its store, not a data path).  A collective on `dist`'s process group: call it on the
    thread that owns that group (not from a watchdog thread that may be abandoned while the main thread uses the group)."""
    uid = [lib.comm_unique_id() if rank == 0 else None]
    if world > 1:
        kw = {}
        if dist.get_backend() == "nccl":        # the pickled id travels as a tensor on this thread's current device
            kw["device"] = torch.device("cuda", torch.cuda.current_device())
        dist.broadcast_object_list(uid, src=0, **kw)
    return uid[0]


def native_comm(lib, rank, world, dist=None, uid=None):
    """An RCCL communicator of our own through the C ABI (include/rt_stereo.h: rt_comm_unique_id / rt_comm_init_rank) -- what a host
    without torch does.  `uid`: the id from exchange_unique_id (otherwise exchanged here).  Call after rt_set_device."""
    if uid is None:
        uid = exchange_unique_id(lib, rank, world, dist)
    return lib.comm_init_rank(world, rank, uid)


def create_nets_native(lib, model, width, height, n_contexts, blob, rank, world, dist=None, max_batch=1, fp16_weights=False, max_disp=0, uid=None):
    """bench.py's multi-GPU start-up on the native entry: rank 0 passes the weight-file image, rt_net_create_broadcast ships it over
    RCCL (ncclBroadcast) and builds the first engine of every rank; the rank's other contexts are built from the image that arrived.
    Returns (nets, crc32 of the image on this rank)."""
    comm = native_comm(lib, rank, world, dist, uid)
    try:
        first = lib.create_broadcast(model, width, height, comm, 0, max_batch=max_batch, blob=blob if rank == 0 else None,
                                     fp16_weights=fp16_weights, max_disp=max_disp)
    finally:
        lib.comm_destroy(comm)
    image = first.weights_image()
    nets = [first] + [lib.create(model, width, height, max_batch=max_batch, weights=image, fp16_weights=fp16_weights, max_disp=max_disp)
                      for _ in range(n_contexts - 1)]
    return nets, first.weights_crc32()
