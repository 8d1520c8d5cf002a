"""Does a captured hipGraph run two independent branches concurrently?  Chain of small dependent GEMMs (M = 32 rows, the decode
projections) on one stream vs the same work split over two forked streams, eager and captured."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L

dev = "cuda"
M, d = 32, 768
g = torch.Generator(device=dev); g.manual_seed(0)
W = [(torch.randn(d, d, device=dev, generator=g) * 0.03).to(torch.bfloat16) for _ in range(8)]
xs = [[torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)] for _ in range(2)]
N = 200


def chain(x, n):
    a, b = x
    for i in range(n):
        L.gemm(a, W[i % 8], b, M, d, d, decode=True)
        a, b = b, a


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one_stream():
    chain(xs[0], N); chain(xs[1], N)


def two_streams():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        chain(xs[0], N)
    with torch.cuda.stream(s2):
        chain(xs[1], N)
    main.wait_stream(s1); main.wait_stream(s2)


def wall(f, reps=5):
    f(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]


print(f"eager   one stream {wall(one_stream):7.2f} ms   two streams {wall(two_streams):7.2f} ms   ({2 * N} launches)")
cs = torch.cuda.Stream()
graphs = {}
for name, f in (("one", one_stream), ("two", two_streams)):
    with torch.cuda.stream(cs):
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=cs):
            f()
    graphs[name] = gr
print(f"graph   one stream {wall(graphs['one'].replay):7.2f} ms   two streams {wall(graphs['two'].replay):7.2f} ms")
