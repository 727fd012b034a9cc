"""N > 1 path on CPU: world_size-2 and -3 gloo process groups shard a frame into row bands, every rank renders
its band (here with the oracle, since there is no GPU), rank 0 gathers them; the assembled frame must be
byte-identical to the single-process frame.  This is the same BandGather / band_of code bench.py runs over
RCCL on the GPU box."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, kind, out_path):
    for sub in ("oracle", "cpp-raytracer-rasterizer_amd"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirt.sharding import BandGather
    from mirt_oracle import Oracle, DEFAULT_LIGHT
    o = Oracle()
    tris = o.soup(9, 200, 0.3) if kind == "soup" else o.cornell()
    g = BandGather(H, W, torch.device("cpu"))
    rot = o.rot_from_yaw(0.2, 1.0)
    r = o.raytrace(tris, (0, 0, -2), rot, H / 2.0, W, H, DEFAULT_LIGHT, y0=g.y0, y1=g.y1, threads=2, want=("xrgb",))
    g.band[: g.y1 - g.y0] = torch.from_numpy(r["xrgb"][g.y0:g.y1].view(np.int32))
    frame = g.gather()
    if rank == 0:
        np.save(out_path, frame.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,kind", [(2, 64, "cornell"), (3, 50, "soup"), (2, 31, "soup")])
def test_band_gather_matches_single_process(tmp_path, oracle, world, H, kind):
    from mirt_oracle import DEFAULT_LIGHT
    W = 48
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, W, H, kind, out), nprocs=world, join=True)
    got = np.load(out)
    tris = oracle.soup(9, 200, 0.3) if kind == "soup" else oracle.cornell()
    ref = oracle.raytrace(tris, (0, 0, -2), oracle.rot_from_yaw(0.2, 1.0), H / 2.0, W, H, DEFAULT_LIGHT, want=("xrgb",))["xrgb"]
    assert np.array_equal(got, ref)


def test_band_of_partitions_rows():
    from mirt.sharding import all_bands
    for H in (1, 7, 540, 1080, 4320):
        for world in (1, 2, 3, 4, 8):
            bands = all_bands(world, H)
            assert bands[0][0] == 0 and bands[-1][1] == H
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in bands]
            assert max(sizes) - min(sizes) <= 1


def _worker_batched(rank, world, port, W, H, batch, out_path):
    for sub in ("oracle", "cpp-raytracer-rasterizer_amd"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirt.sharding import BandGather
    from mirt_oracle import Oracle, DEFAULT_LIGHT
    o = Oracle()
    tris = o.cornell()
    g = BandGather(H, W, torch.device("cpu"), depth=2, batch=batch)
    for which in (0, 1):                                   # both band buffers, every frame of a batch from another camera yaw
        for b in range(batch):
            rot = o.rot_from_yaw(0.1 * (b + 1) + which, 1.0)
            r = o.raytrace(tris, (0, 0, -2), rot, H / 2.0, W, H, DEFAULT_LIGHT, y0=g.y0, y1=g.y1, threads=2, want=("xrgb",))
            g.slot(which, b)[: g.y1 - g.y0] = torch.from_numpy(r["xrgb"][g.y0:g.y1].view(np.int32))
        g.gather(which)
        if rank == 0:
            np.save(out_path % which, g.frames.numpy().view(np.uint32).copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,batch", [(2, 40, 3), (3, 31, 4)])
def test_batched_band_gather(tmp_path, oracle, world, H, batch):
    """Several frames' bands per collective (what bench.py does for frames that render faster than a collective starts):
    every frame of the batch must come out whole, for even and uneven bands and for both band buffers."""
    from mirt_oracle import DEFAULT_LIGHT
    W = 40
    out = str(tmp_path / "frames%d.npy")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_batched, args=(world, port, W, H, batch, out), nprocs=world, join=True)
    tris = oracle.cornell()
    for which in (0, 1):
        got = np.load(out % which)
        assert got.shape == (batch, H, W)
        for b in range(batch):
            rot = oracle.rot_from_yaw(0.1 * (b + 1) + which, 1.0)
            ref = oracle.raytrace(tris, (0, 0, -2), rot, H / 2.0, W, H, DEFAULT_LIGHT, want=("xrgb",))["xrgb"]
            assert np.array_equal(got[b], ref)


def _worker_strips(rank, world, port, W, H, strip, root, out_path):
    for sub in ("oracle", "cpp-raytracer-rasterizer_amd"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mirt
    from mirt_oracle import Oracle, DEFAULT_LIGHT
    o = Oracle()
    tris = o.soup(9, 200, 0.3)
    rot = o.rot_from_yaw(0.2, 1.0)
    # what mirt_*_sharded does with strip_rows > 0: this rank's strips into its band buffer back to back (the root's in place), then
    # the pieces of mirt_partition_plan as point-to-point messages to the root
    segs = mirt.partition_segments(rank, world, H, strip)
    band = np.zeros((sum(b - a for a, b in segs), W), np.uint32)
    at = 0
    for a, b in segs:
        band[at:at + b - a] = o.raytrace(tris, (0, 0, -2), rot, H / 2.0, W, H, DEFAULT_LIGHT, y0=a, y1=b, threads=2, want=("xrgb",))["xrgb"][a:b]
        at += b - a
    plan = mirt.partition_plan(world, root, W, H, 1, strip)
    if rank == root:
        frame = np.zeros((H, W), np.uint32)
        at = 0
        for a, b in segs:
            frame[a:b] = band[at:at + b - a]
            at += b - a
        flat = frame.view(np.uint8).reshape(-1)
        for ro, bo, nbytes, peer in plan:
            t = torch.empty(nbytes, dtype=torch.uint8)
            dist.recv(t, src=peer)
            flat[ro:ro + nbytes] = t.numpy()
        np.save(out_path, frame)
    else:
        flat = band.view(np.uint8).reshape(-1)
        for ro, bo, nbytes, peer in plan:
            if peer == rank:
                dist.send(torch.from_numpy(flat[bo:bo + nbytes].copy()), dst=root)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,strip,root", [(2, 64, 8, 0), (3, 50, 16, 1)])
def test_strip_partition_gather_matches_single_process(tmp_path, oracle, world, H, strip, root):
    """Interleaved strips (mirt_set_partition): every rank renders its strips, the plan's pieces travel to the root as
    point-to-point messages of a gloo group, and the assembled frame equals the single-process frame."""
    from mirt_oracle import DEFAULT_LIGHT
    W = 40
    out = str(tmp_path / "frame.npy")
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_strips, args=(world, port, W, H, strip, root, out), nprocs=world, join=True)
    got = np.load(out)
    ref = oracle.raytrace(oracle.soup(9, 200, 0.3), (0, 0, -2), oracle.rot_from_yaw(0.2, 1.0), H / 2.0, W, H, DEFAULT_LIGHT, want=("xrgb",))["xrgb"]
    assert np.array_equal(got, ref)


def _worker_weighted(rank, world, port, W, H, root, out_path):
    for sub in ("oracle", "cpp-raytracer-rasterizer_amd"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mirt
    from mirt_oracle import Oracle, DEFAULT_LIGHT
    o = Oracle()
    tris = o.soup(9, 200, 0.3)
    rot = o.rot_from_yaw(0.2, 1.0)
    # What mirt_*_sharded does under MIRT_PARTITION_WEIGHTED: every rank derives the SAME boundaries from the same cost histogram
    # by integer arithmetic (mirt_weighted_bounds) -- here a histogram with its weight in the lower third of the frame, which every
    # rank computes for itself, as the ranks of a sharded frame do -- renders its band and sends it where mirt_bounds_plan says.
    tile_rows = (H + 7) // 8
    hist = (np.arange(tile_rows) >= 2 * tile_rows // 3).astype(np.uint32) * np.uint32(50000) + np.uint32(100)
    bounds = mirt.weighted_bounds(hist, 0, W, H, world)
    everyone = [None] * world
    dist.all_gather_object(everyone, bounds)               # (the test's own check that no rank disagrees; the product exchanges nothing)
    assert all(b == bounds for b in everyone)
    a, b = bounds[rank], bounds[rank + 1]
    band = o.raytrace(tris, (0, 0, -2), rot, H / 2.0, W, H, DEFAULT_LIGHT, y0=a, y1=b, threads=2, want=("xrgb",))["xrgb"][a:b].copy()
    plan = mirt.bounds_plan(world, root, W, H, 1, bounds)
    if rank == root:
        frame = np.zeros((H, W), np.uint32)
        frame[a:b] = band
        flat = frame.view(np.uint8).reshape(-1)
        for ro, bo, nbytes, peer in plan:
            t = torch.empty(nbytes, dtype=torch.uint8)
            dist.recv(t, src=peer)
            flat[ro:ro + nbytes] = t.numpy()
        np.save(out_path, frame)
        np.save(out_path + ".bounds.npy", np.array(bounds))
    else:
        flat = band.view(np.uint8).reshape(-1)
        for ro, bo, nbytes, peer in plan:
            if peer == rank:
                dist.send(torch.from_numpy(flat[bo:bo + nbytes].copy()), dst=root)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,root", [(2, 64, 0), (3, 96, 1)])
def test_weighted_partition_gather_matches_single_process(tmp_path, oracle, world, H, root):
    """Bands of equal estimated cost (MIRT_PARTITION_WEIGHTED): every rank computes the boundaries itself, the plan's pieces
    travel to the root as point-to-point messages of a gloo group, and the assembled frame equals the single-process frame."""
    from mirt_oracle import DEFAULT_LIGHT
    W = 40
    out = str(tmp_path / "frame.npy")
    port = 35500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_weighted, args=(world, port, W, H, root, out), nprocs=world, join=True)
    got = np.load(out)
    bounds = np.load(out + ".bounds.npy")
    sizes = np.diff(bounds)
    assert sizes[-1] < sizes[0]                             # the heavy lower third makes the last band the shortest
    ref = oracle.raytrace(oracle.soup(9, 200, 0.3), (0, 0, -2), oracle.rot_from_yaw(0.2, 1.0), H / 2.0, W, H, DEFAULT_LIGHT, want=("xrgb",))["xrgb"]
    assert np.array_equal(got, ref)
