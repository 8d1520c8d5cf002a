"""Where the ~5.6 us of a decode-step skinny GEMM go when it is one link of a dependent chain: a SKINNY_STAMPS build of the library
(tools/libvid2seq_hip_stamps.so: -DSKINNY_STAMPS=1) lets thread 0 of the first and of the last block stamp s_memrealtime (100 MHz, chip-wide) at the
kernel's phase boundaries into the workspace of each launch; this script replays a graph of dependent QKV / O launches (as
tools/skinny_chain_probe.py) and prints, averaged over the chain: previous launch's last stamp -> this launch's entry (the boundary), and
the phases inside (arguments + addresses, K loop until every load has arrived, partials to LDS, barrier, reduction, epilogue + store).
usage: V2S_LIB=tools/libvid2seq_hip_stamps.so python tools/skinny_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

dev = torch.device("cuda")
torch.manual_seed(0)
NW, n = 120, 240
x0 = torch.randn(64, 768, device=dev).to(torch.bfloat16)
Wqkv = [torch.randn(2304, 768, device=dev).to(torch.bfloat16) * 0.03 for _ in range(NW)]
Wo = [torch.randn(768, 768, device=dev).to(torch.bfloat16) * 0.03 for _ in range(NW)]
qkv = torch.zeros(64, 2304, device=dev, dtype=torch.bfloat16)
xa, xb = x0.clone(), x0.clone()
WSF = 2 * (32 + 2 * 1024)            # floats per launch: 32 u64 of phase stamps + (entry, exit) of up to 1024 blocks
ws = torch.zeros(2 * n, WSF, device=dev, dtype=torch.float32)


def body(cold):
    a, b = xa, xb
    for i in range(n):
        j = i % NW if cold else 0
        L.gemm(a, Wqkv[j], qkv, 64, 2304, 768, rms_eps=1e-6, decode=True, workspace=ws[2 * i])
        L.gemm(qkv, Wo[j], b, 64, 768, 768, lda=2304, residual=a, decode=True, workspace=ws[2 * i + 1])
        a, b = b, a


names = ["args+addr", "K loop (loads landed)", "partials -> LDS", "barrier", "reduce 8 partials", "epilogue + store"]
for cold in (0, 1):
    body(cold); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(cold)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1000 / (2 * n)
    full = ws.view(torch.int64).cpu().view(2 * n, WSF // 2)
    st = full[:, :32]                   # [launch][first block: cycles 0..7, ref 8..15 | last block: 16..23, 24..31]
    print(f"--- weights {'rotating (HBM)' if cold else 'hot'}: {per:.2f} us per launch node to node")
    for which, name in ((0, "QKV 64x2304x768 (576 blocks)"), (1, "O 64x768x768 + residual (192 blocks)")):
        rows = st[which::2][4:]                                        # skip the first launches of the replay
        prev = st[(1 - which)::2]
        prev = prev[4:] if which == 1 else prev[3:-1]                  # the launch before: O follows QKV of the same i, QKV follows O of i - 1
        ref = lambda r, blk, i: r[:, blk * 16 + 8 + i].double() * 0.01          # 100 MHz ticks -> us
        cyc = lambda r, blk, i: r[:, blk * 16 + i].double()
        span_first = (ref(rows, 0, 6) - ref(rows, 0, 0)).mean()
        span_all = (torch.maximum(ref(rows, 0, 6), ref(rows, 1, 6)) - torch.minimum(ref(rows, 0, 0), ref(rows, 1, 0))).mean()
        gap = (torch.minimum(ref(rows, 0, 0), ref(rows, 1, 0)) - torch.maximum(ref(prev, 0, 6), ref(prev, 1, 6))).mean()
        last_entry = (ref(rows, 1, 0) - ref(rows, 0, 0)).mean()
        print(f"  {name}: previous launch's last stamp -> first entry {gap:.2f} us; first block entry -> store {span_first:.2f} us; last block enters {last_entry:+.2f} us after the first; "
              f"first entry -> last store {span_all:.2f} us")
        nb = 576 if which == 0 else 192
        ent = full[which::2][4:, 32:32 + 2 * nb:2].double() * 0.01
        ext = full[which::2][4:, 33:33 + 2 * nb:2].double() * 0.01
        e_sorted = (ent - ent.min(1, keepdim=True).values).sort(1).values.mean(0)
        print("    entry of the n-th block after the first (us): " + "  ".join(f"#{i + 1} {e_sorted[i]:.2f}" for i in (63, 127, 191, 255, 319, 383, 447, 511, 575) if i < nb)
              + f";  blocks alive per block {(ext - ent).mean():.2f} us;  first entry -> last exit {(ext.max(1).values - ent.min(1).values).mean():.2f} us")
        for blk, bn in ((0, "first block"), (1, "last block ")):
            d = [(ref(rows, blk, i + 1) - ref(rows, blk, i)).mean() for i in range(6)]
            print(f"    {bn}: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, d)) + "  us")
