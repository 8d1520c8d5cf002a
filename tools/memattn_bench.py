"""Decode-step cross-attention: K/V-cache kernel (12 layers = 12 different K|V tensors) against the shared-memory formulation
(qfold + memattn + ctxfold, 12 layers read the SAME memory).  usage: python tools/memattn_bench.py [B] [G] [S]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
dev = torch.device("cuda", 0)
H, d, W, NL = 12, 768, 768, 12
g = torch.Generator().manual_seed(0)
klen_l = [100 + int(x) for x in torch.randint(600, 1001, (B,), generator=g)] if S >= 1100 else [S] * B
klen = torch.tensor(klen_l, dtype=torch.int32, device=dev)
mask = (torch.arange(S, device=dev)[None, :] < klen[:, None]).to(torch.uint8).contiguous()
rows = B * G
mem = torch.randn(B, S, d, device=dev).bfloat16()
kv = [torch.randn(B, S, 2 * W, device=dev).bfloat16() for _ in range(NL)]
q = torch.randn(rows, W, device=dev).bfloat16() * 0.3
x = torch.randn(rows, d, device=dev).bfloat16()
wq = [(torch.randn(W, d, device=dev) * 0.01).bfloat16() for _ in range(NL)]
wkT = [(torch.randn(d, W, device=dev) * 0.03).bfloat16() for _ in range(NL)]
wv = [(torch.randn(W, d, device=dev) * 0.03).bfloat16() for _ in range(NL)]
ctx = torch.empty(rows, W, dtype=torch.bfloat16, device=dev)
qp = torch.empty(rows, H, d, dtype=torch.bfloat16, device=dev)
nt = (max(klen_l) + 31) // 32

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / NL          # us per layer

def old():
    for i in range(NL):
        L.decode_attn(rows, H, S, q, W, kv[i], kv[i][:, :, W:], S * 2 * W, 2 * W, ctx, W, key_mask=mask, mask_ld=S, kv_group=G if G > 1 else 0)
print(f"B={B} G={G} S={S}  valid keys {sum(klen_l)}  K|V bytes/layer {sum(klen_l) * 2 * W * 2 / 1e6:.1f} MB, memory bytes {sum(klen_l) * d * 2 / 1e6:.1f} MB")
t = timeit(old)
print(f"K/V cache kernel                 {t:7.2f} us/layer  ({sum(klen_l) * 2 * W * 2 / t / 1e6:.2f} TB/s of valid K|V)")
for tpp in (18, 9, 5):
    plan = L.MemAttnPlan(klen_l, G * H, dev, tiles_per_piece=tpp)
    def qf():
        for i in range(NL):
            L.decode_qfold(x, rows, wq[i], wkT[i], 1e-6, qp, H, d)
    def ma():
        for i in range(NL):
            L.decode_memattn(qp, mem, S * d, plan, d)
    def cf():
        for i in range(NL):
            L.decode_ctxfold(plan, rows, G, H, wv[i], ctx, d)
    def allk():
        for i in range(NL):
            L.decode_qfold(x, rows, wq[i], wkT[i], 1e-6, qp, H, d)
            L.decode_memattn(qp, mem, S * d, plan, d)
            L.decode_ctxfold(plan, rows, G, H, wv[i], ctx, d)
    a, b, c, al = timeit(qf), timeit(ma), timeit(cf), timeit(allk)
    tiles = [int(t >> 16) - int(t & 0xffff) for t in plan.blk_host[:, 1]]
    print(f"blocks {plan.nblk:3d} (tiles/block {min(tiles)}-{max(tiles)}): qfold {a:6.2f}  memattn {b:6.2f} ({sum(klen_l) * d * 2 / b / 1e6:.2f} TB/s)  ctxfold {c:6.2f}  chained {al:6.2f} us/layer")
