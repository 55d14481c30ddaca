"""CPU, world_size 2 over gloo: the N>1 host path - contiguous batch sharding and the single
all-gather of (B_rank, K, 33) detections - equals the unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, equal_hint, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from centerfusiondetect3d_amd.distributed import shard_range, gather_detections, assume_equal_shards
    assume_equal_shards(equal_hint)
    full = torch.arange(n_frames * 100 * 33, dtype=torch.float32).view(n_frames, 100, 33)
    lo, hi = shard_range(n_frames, rank, world)
    out = gather_detections(full[lo:hi].clone())
    q.put((rank, lo, hi, bool(torch.equal(out, full)), tuple(out.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,equal_hint", [(8, False), (8, True), (7, False), (1, False)])
def test_gather_detections_gloo_ws2(n_frames, equal_hint):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, equal_hint, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_frames   # contiguous cover
    for _, _, _, same, shape in res:
        assert same and shape == (n_frames, 100, 33)


def _worker_async(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from centerfusiondetect3d_amd.distributed import DetectionGatherer, assume_equal_shards
    assume_equal_shards(True)
    g = DetectionGatherer()
    ok = True
    pending = []
    for step in range(3):                                    # three steps in flight before the first wait
        mine = torch.full((4, 100, 54), float(10 * step + rank))
        pending.append((step, g.submit(mine)))
    for step, h in pending:
        out = h.wait()
        exp = torch.cat([torch.full((4, 100, 54), float(10 * step + r)) for r in range(world)], 0)
        ok = ok and bool(torch.equal(out, exp))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_gatherer_gloo_ws2():
    """DetectionGatherer (async all-gather of the final (B,K,54) rows, several steps in flight)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res == [(0, True), (1, True)]


def test_shard_range_balanced():
    from centerfusiondetect3d_amd.distributed import shard_range
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    from centerfusiondetect3d_amd.distributed import gather_detections
    t = torch.randn(3, 100, 33)
    assert gather_detections(t) is t
