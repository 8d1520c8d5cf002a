"""One shot of the decode cross-attention kernels for the PMC traffic passes (tools/pmc_memattn.sh): 12 launches each of the K/V-cache
kernel (12 different K|V tensors = 12 layers) and of the memory pass (the same memory 12 times), B = 64, 100 frames + 600-1000 tokens."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = torch.device("cuda", 0)
B, S, H, d, W, NL = 64, 1100, 12, 768, 768, 12
g = torch.Generator().manual_seed(0)
klen_l = [100 + int(x) for x in torch.randint(600, 1001, (B,), generator=g)]
klen = torch.tensor(klen_l, dtype=torch.int32, device=dev)
mask = (torch.arange(S, device=dev)[None, :] < klen[:, None]).to(torch.uint8).contiguous()
mem = torch.randn(B, S, d, device=dev).bfloat16()
kv = [torch.randn(B, S, 2 * W, device=dev).bfloat16() for _ in range(NL)]
q = torch.randn(B, W, device=dev).bfloat16() * 0.3
qp = torch.randn(B, H, d, device=dev).bfloat16() * 0.05
ctx = torch.empty(B, W, dtype=torch.bfloat16, device=dev)
plan = L.MemAttnPlan(klen_l, H, dev)
for rep in range(2):
    for i in range(NL):
        L.decode_attn(B, H, S, q, W, kv[i], kv[i][:, :, W:], S * 2 * W, 2 * W, ctx, W, key_mask=mask, mask_ld=S)
    for i in range(NL):
        L.decode_memattn(qp, mem, S * d, plan, d)
torch.cuda.synchronize()
print("valid keys", sum(klen_l), "K|V bytes/layer", sum(klen_l) * 2 * W * 2, "memory bytes/layer", sum(klen_l) * d * 2)
