"""Would two half-size wavefront loops on two streams overlap usefully?  Two contexts (a stream each), two scenes, two host
threads, each rendering half of the bench's samples (sample_first/stride shards) at the same time, against one render of all
of them."""
import importlib, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
flat = hm.cbox_scene(512, 512)
ctxs = [abi.Context(0), abi.Context(0)]
scs = [abi.Scene(c, flat) for c in ctxs]
films = [torch.zeros((512, 512, 5), dtype=torch.float32, device="cuda") for _ in range(2)]
full = abi.render_params(spp=512)
halves = [abi.render_params(spp=512, sample_first=r, sample_stride=2) for r in range(2)]
def one():
    scs[0].render_device(full, films[0].data_ptr())
def two():
    ts = [threading.Thread(target=lambda r=r: scs[r].render_device(halves[r], films[r].data_ptr())) for r in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
def wall(f, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
one(); two(); two()
a, b, a2, b2 = wall(one), wall(two), wall(one), wall(two)
print("one render of 512 spp: %.2f / %.2f ms; two concurrent renders of 256 spp each: %.2f / %.2f ms" % (a, a2, b, b2))
