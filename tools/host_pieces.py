"""Developer aid: host cost of the pieces of one fresh-feed pass (stream context, pinned H2D enqueue, event create / record, graph launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
host = torch.empty(70000, dtype=torch.uint8).pin_memory()
d = torch.empty(70000, dtype=torch.uint8, device=dev)
N = 2000

def t(name, f):
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(N):
        f()
    b = time.perf_counter()
    torch.cuda.synchronize()
    print("%-44s %7.2f us" % (name, (b - a) / N * 1e6))

def ctx():
    with torch.cuda.stream(st):
        pass
t("with torch.cuda.stream: pass", ctx)
def cp():
    with torch.cuda.stream(st):
        d.copy_(host, non_blocking=True)
t("ctx + copy_(pinned, non_blocking)", cp)
def evn():
    ev = torch.cuda.Event()
    ev.record(st)
t("Event() + record(stream)", evn)
ev0 = torch.cuda.Event()
t("record(stream) on an existing event", lambda: ev0.record(st))
t("event.synchronize() (already done)", lambda: ev0.synchronize())
from fcl_taco2_amd import _lib
lib = _lib.load()
if hasattr(lib, "fcl_h2d_async"):
    sp, dp, h = host.data_ptr(), d.data_ptr(), st.cuda_stream
    t("fcl_h2d_async (ctypes)", lambda: lib.fcl_h2d_async(dp, sp, 70000, h))
g = torch.cuda.CUDAGraph()
x = torch.zeros(1024, device=dev)
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        for _ in range(90):
            x.add_(1.0)
def rp():
    with torch.cuda.stream(st):
        g.replay()
t("ctx + replay of a 90-node graph", rp)
