"""Dev (GPU box): what happens when a captured graph that has become cyclic garbage is collected INSIDE another stream
capture ("a crash found by the full suite in round 3").  Plain torch, no model.
    python tools/repro_gc_capture.py graph|event|tensor"""
import gc, sys
import torch

what = sys.argv[1] if len(sys.argv) > 1 else "graph"
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()


class Holder:
    pass


def make_garbage():
    h = Holder()
    h.me = h                                   # a cycle: only the collector frees it
    if what == "graph":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            h.y = x + 1
        h.g = g
    elif what == "event":
        s = torch.cuda.Stream()
        e = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            h.y = x + 1
            e.record(s)
        h.e, h.s = e, s
    else:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            h.y = torch.ones(1 << 20, device="cuda")
        h.y.record_stream(torch.cuda.current_stream())


gc.collect()
gc.disable()
make_garbage()
g2 = torch.cuda.CUDAGraph()
print("capturing with pending garbage:", what, flush=True)
with torch.cuda.graph(g2):
    z = x * 2
    n = gc.collect()                           # the collection happens while the stream is capturing
    z = z + 1
print("survived; collected", n, flush=True)
g2.replay()
torch.cuda.synchronize()
print("replayed", float(z[0]), flush=True)
