"""Dev experiment: does running the MFMA-bound head launches of one sub-batch UNDER the latency-bound trunk of another
pay?  Two host threads drive one model on two HIP streams, each a full forward + decode of half a batch (plans are
keyed per stream), the second thread offset by `stagger` ms so its trunk meets the other's heads.
    python tools/bench_stagger.py [batch=16] [iters=40]"""
import os, sys, time, threading, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_post_packed
H, W = 448, 800
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda")
model = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W))), seed=0).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(B, H, W, dev, seed=1)
meta = bench.make_meta(B, H, W) if hasattr(bench, "make_meta") else None

def step(img, pc, cal):
    return model(img, pc_dep=pc, calib=cal)

def run_single(streams):
    model.streams = streams
    with torch.no_grad():
        for _ in range(4): step(images, pc_dep, calib)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(K): step(images, pc_dep, calib)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / K

def run_threads(n_thr, stagger_ms):
    model.streams = 1
    shards = [(images[i::n_thr].contiguous(), pc_dep[i::n_thr].contiguous(), calib[i::n_thr].contiguous()) for i in range(n_thr)]
    global _thread_streams
    if "_thread_streams" not in globals():
        _thread_streams = [torch.cuda.Stream() for _ in range(n_thr)]
    streams = _thread_streams
    with torch.no_grad():
        for s, sh in zip(streams, shards):
            with torch.cuda.stream(s):
                for _ in range(3): step(*sh)
    torch.cuda.synchronize()
    start = threading.Barrier(n_thr + 1)
    def worker(i):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            start.wait()
            if i: time.sleep(i * stagger_ms * 1e-3)
            for _ in range(K): step(*shards[i])
            streams[i].synchronize()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_thr)]
    for t_ in th: t_.start()
    start.wait(); t = time.perf_counter()
    for t_ in th: t_.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / K

# (the first streams a process creates get their own hardware queues, later ones share: run the two modes in separate
#  processes so that each measures with "first" streams:  ... 16 40 single   /   ... 16 40 threads)
mode = sys.argv[3] if len(sys.argv) > 3 else "single"
if mode == "single":
    for s in (1, 2):
        dt = run_single(s)
        print(f"one thread, model.streams={s}: {dt * 1e3:.3f} ms per {B} frames = {B / dt:.0f} frames/s", flush=True)
else:
    for stg in (0.0, 1.0, 2.0, 3.0, 4.0):
        dt = run_threads(2, stg)
        print(f"2 threads x {B // 2} frames, stagger {stg} ms: {dt * 1e3:.3f} ms per {B} frames = {B / dt:.0f} frames/s", flush=True)
