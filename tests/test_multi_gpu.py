"""N > 1 path on CPU: two processes, gloo backend.  Rank 0 broadcasts the weight-file image
(redtail_amd/parallel.py, the same code bench.py runs over RCCL), every rank builds its own engine from
it, processes its shard of the stereo pairs (emulator backend) and the gathered result is checked
against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from redtail_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, w, h, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from redtail_amd import build, capi, synth
    blob = capi.pack_weights(synth.synth_weights_resnet18_2d()) if rank == 0 else b""
    blob = parallel.broadcast_blob(blob, rank, torch.device("cpu"), dist)
    lib = capi.NetLib(build.build_host_emu(), build.build_emu())
    net = lib.create("resnet18_2D", w, h, max_batch=1, weights=blob, max_disp=6)
    mine = list(parallel.shard(n_pairs, rank, world))
    out = np.zeros((n_pairs, 1, h, w), np.float32)
    for i in mine:
        l, r = synth.synth_pair(h, w, 1234 + i)
        o = np.full((1, 1, h, w), np.nan, np.float32)
        net.execute(l[None].copy(), r[None].copy(), o, 1)
        out[i] = o[0]
    t = torch.from_numpy(out)
    dist.all_reduce(t)                     # test-side gather of the per-rank shards (zeros elsewhere)
    if rank == 0:
        np.save(os.path.join(tmp, "out.npy"), t.numpy())
        np.save(os.path.join(tmp, "blob_len.npy"), np.array([len(blob)]))
    dist.destroy_process_group()


def test_shard_partition():
    for n in (0, 1, 5, 8, 64):
        for world in (1, 2, 3, 8):
            parts = [list(parallel.shard(n, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_two_rank_broadcast_and_sharding(tmp_path):
    from oracle import stereo_oracle as O
    from redtail_amd import build, synth
    build.build_host_emu()                 # build once, before forking
    n_pairs, w, h = 3, 33, 17
    mp.spawn(_worker, args=(2, _free_port(), n_pairs, w, h, str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npy")
    weights = synth.synth_weights_resnet18_2d()
    ls, rs = zip(*(synth.synth_pair(h, w, 1234 + i) for i in range(n_pairs)))
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(np.stack(ls)), torch.from_numpy(np.stack(rs)), weights, max_disp=6).numpy()
    assert np.abs(out - ref).max() <= 1e-3
