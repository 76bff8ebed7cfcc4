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


def test_native_broadcast_entry_world_of_one_on_emulator():
    """rt_comm_* / rt_net_create_broadcast / rt_net_weights_crc32 / rt_net_weights_image through ctypes on the emulator build (which has no
    RCCL: a world of one rank): same engine, same crc32 as zlib, and the image comes back byte for byte"""
    import zlib
    from oracle import stereo_oracle as O
    from redtail_amd import build, capi, synth
    lib = capi.NetLib(build.build_host_emu(), build.build_emu())
    weights = synth.synth_weights_resnet18_2d()
    blob = capi.pack_weights(weights)
    nets, crc = parallel.create_nets_native(lib, "resnet18_2D", 33, 17, 2, blob, 0, 1, max_disp=6)
    assert crc == zlib.crc32(blob) & 0xffffffff
    assert nets[0].weights_image() == blob
    l, r = synth.synth_pair(17, 33, 7)
    outs = []
    for net in nets:
        o = np.full((1, 1, 17, 33), np.nan, np.float32)
        net.execute(l[None].copy(), r[None].copy(), o, 1)
        outs.append(o)
        net.destroy()
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], weights, max_disp=6).numpy()
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0] - ref).max() <= 1e-3
    with pytest.raises(capi.RtError):
        lib.comm_init_rank(2, 0, b"\0" * 128)            # the emulator build refuses a world it cannot serve


def test_native_broadcast_entry_two_threads_on_emulator():
    """The threading model of apps/stereo_throughput.cpp -- one process, one thread per device, rt_comm_init_all (ncclCommInitAll), every
    thread calls rt_net_create_broadcast on its own communicator -- on the emulator build, whose rt_comm_init_all returns communicators
    joined by an in-process stand-in for RCCL (a real collective: every rank must call, the root's bytes reach all).  Two ranks: rank 1
    passes no image and receives rank 0's; both engines hold the same bytes (crc32), answer like the oracle, and a rank that skips the
    collective would hang -- so the test also bounds the time."""
    import threading
    import zlib
    from oracle import stereo_oracle as O
    from redtail_amd import build, capi, synth
    lib = capi.NetLib(build.build_host_emu(), build.build_emu())
    weights = synth.synth_weights_resnet18_2d()
    blob = capi.pack_weights(weights)
    comms = lib.comm_init_all(2)
    nets, errors = [None, None], []

    def rank_main(rank):
        try:
            nets[rank] = lib.create_broadcast("resnet18_2D", 33, 17, comms[rank], 0, blob=blob if rank == 0 else None, max_disp=6)
        except Exception as e:                     # noqa: BLE001 -- reported by the main thread
            errors.append((rank, e))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in (1, 0)]      # the receiving rank first
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "the broadcast start-up did not finish: a rank is stuck in the collective"
    assert not errors, errors
    crc = zlib.crc32(blob) & 0xffffffff
    assert nets[0].weights_crc32() == crc and nets[1].weights_crc32() == crc
    assert nets[1].weights_image() == blob
    l, r = synth.synth_pair(17, 33, 7)
    outs = []
    for net in nets:                               # (the emulator runs one kernel at a time: the passes are sequential)
        o = np.full((1, 1, 17, 33), np.nan, np.float32)
        net.execute(l[None].copy(), r[None].copy(), o, 1)
        outs.append(o)
        net.destroy()
    for c in comms:
        lib.comm_destroy(c)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], weights, max_disp=6).numpy()
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0] - ref).max() <= 1e-3


# ---- GPU tier: the same code over RCCL (torch.distributed backend "nccl") ------------------------------------------------
def _native_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)      # only carries the 128-byte unique id
    from redtail_amd import capi, model_files, synth
    lib = capi.NetLib()
    lib.kernels.check(lib.kernels.lib.rt_set_device(rank), "rt_set_device")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    blob = open(model_files.weight_file("resnet18_2D"), "rb").read() if rank == 0 else None
    w, h = 257, 129
    nets, crc = parallel.create_nets_native(lib, "resnet18_2D", w, h, 2, blob, rank, world, dist)
    l, r = synth.synth_pair(h, w, 1234 + rank)
    out = torch.full((1, 1, h, w), float("nan"), device=dev)
    nets[1].execute(torch.from_numpy(l)[None].to(dev), torch.from_numpy(r)[None].to(dev), out, 1)
    torch.cuda.synchronize(dev)
    np.save(os.path.join(tmp, "nout%d.npy" % rank), out.cpu().numpy())
    np.save(os.path.join(tmp, "ncrc%d.npy" % rank), np.array([crc]))
    for n in nets:
        n.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_native_rccl_entry(tmp_path, world):
    """The native multi-GPU entry (no torch in the data or weight path): rt_comm_unique_id -> rt_comm_init_rank (ncclCommInitRank) ->
    rt_net_create_broadcast (ncclBroadcast of the image) on every rank; crc32 of what arrived == the file's; each rank's second context,
    built from the received image, matches the oracle on the rank's own pair.  World 2 when two devices are visible."""
    import zlib
    from oracle import stereo_oracle as O
    from redtail_amd import capi, model_files, synth
    if torch.cuda.device_count() < world:
        pytest.skip("%d visible device(s)" % torch.cuda.device_count())
    mp.spawn(_native_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    path = model_files.weight_file("resnet18_2D")
    want = zlib.crc32(open(path, "rb").read()) & 0xffffffff
    weights = capi.read_weights(path)
    for rank in range(world):
        assert int(np.load(tmp_path / ("ncrc%d.npy" % rank))[0]) == want
        l, r = synth.synth_pair(129, 257, 1234 + rank)
        with torch.no_grad():
            ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], weights).numpy()
        assert np.abs(np.load(tmp_path / ("nout%d.npy" % rank)) - ref).max() <= 1e-3


@pytest.mark.gpu
def test_native_throughput_app():
    """apps/stereo_throughput.cpp (one thread per device, rt_comm_init_all + rt_net_create_broadcast) on every visible device"""
    import json
    import subprocess
    import zlib
    from redtail_amd import model_files
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    app = os.path.join(ROOT, "redtail_amd", "lib", "stereo_throughput")
    assert os.path.exists(app), "redtail_amd/lib/stereo_throughput not built (__graft_entry__.build())"
    path = model_files.weight_file("resnet18_2D")
    res = subprocess.run([app, "resnet18_2D", "513", "257", path, "--pairs", "48", "--contexts", "3", "--warmup", "6"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == torch.cuda.device_count() and line["value"] > 0
    assert line["weights_crc32"] == "%08x" % (zlib.crc32(open(path, "rb").read()) & 0xffffffff)
    assert all(r["weights_crc32"] == line["weights_crc32"] for r in line["ranks"])



def _rccl_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import zlib
    from redtail_amd import capi, model_files
    blob = open(model_files.weight_file("resnet18_2D"), "rb").read() if rank == 0 else b""
    blob = parallel.broadcast_blob(blob, rank, dev, dist)
    crc = torch.tensor([zlib.crc32(blob) & 0xffffffff], dtype=torch.int64, device=dev)
    crcs = [torch.zeros_like(crc) for _ in range(world)]
    dist.all_gather(crcs, crc)
    # every rank builds its engine from the broadcast image and runs its own pair
    lib = capi.NetLib()
    lib.kernels.check(lib.kernels.lib.rt_set_device(rank), "rt_set_device")
    from redtail_amd import synth
    w, h = 257, 129
    net = lib.create("resnet18_2D", w, h, weights=blob)
    l, r = synth.synth_pair(h, w, 1234 + rank)
    out = torch.full((1, 1, h, w), float("nan"), device=dev)
    net.execute(torch.from_numpy(l)[None].to(dev), torch.from_numpy(r)[None].to(dev), out, 1)
    torch.cuda.synchronize(dev)
    np.save(os.path.join(tmp, "out%d.npy" % rank), out.cpu().numpy())
    if rank == 0:
        np.save(os.path.join(tmp, "crcs.npy"), np.array([int(c.item()) for c in crcs]))
        np.save(os.path.join(tmp, "world.npy"), np.array([dist.get_world_size()]))
    net.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_rccl_broadcast_of_weights(tmp_path, world):
    """bench.py's multi-GPU start-up over the real backend: rank 0 broadcasts the weight-file image with RCCL, every rank
    checks it (crc32 all-gathered), builds its engine from it on ITS device and matches the oracle on its own pair.
    World size 1 always runs; world size 2 when two devices are visible (the driver's multi-GPU node)."""
    import zlib
    from oracle import stereo_oracle as O
    from redtail_amd import capi, model_files, synth
    if torch.cuda.device_count() < world:
        pytest.skip("%d visible device(s)" % torch.cuda.device_count())
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    path = model_files.weight_file("resnet18_2D")
    want = zlib.crc32(open(path, "rb").read()) & 0xffffffff
    assert list(np.load(tmp_path / "crcs.npy")) == [want] * world
    assert int(np.load(tmp_path / "world.npy")[0]) == world
    weights = capi.read_weights(path)
    for rank in range(world):
        l, r = synth.synth_pair(129, 257, 1234 + rank)
        with torch.no_grad():
            ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], weights).numpy()
        assert np.abs(np.load(tmp_path / ("out%d.npy" % rank)) - ref).max() <= 1e-3


@pytest.mark.gpu
def test_bench_multi_rank_startup_with_a_world_of_one():
    """bench.py itself under the driver's launcher (torch.distributed.run), with BENCH_FORCE_DISTRIBUTED=1 so that a single rank
    walks the multi-rank start-up: process group on RCCL, native entry (rt_comm_init_rank + rt_net_create_broadcast) inside its
    watchdog thread, agreement on the transport, barrier + max over ranks.  The JSON line names the transport."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_DISTRIBUTED="1")
    env.pop("RT_DEV_KNOBS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3",
           "--no-secondary", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["weights_transport"].startswith("rccl-native"), (line["config"]["weights_transport"], out.stderr[-1500:])
    assert "weight image crc32" in out.stderr
