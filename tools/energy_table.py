"""Energy per launch of the train step's kernel classes ("if energy-bound, rank changes by energy, not by isolated us").
Each kernel runs alone, back to back without host syncs, for ~0.4-1 s; time per launch (HIP events), effective shader clock
(v2s_clock_probe), socket power (amdgpu hwmon, host thread) -> joules per launch = W x s, and whether the launch is at the 1400 W cap
(power-bound: only its ENERGY counts inside the step) or below it (time-bound: its duration counts).
usage: python tools/energy_table.py [--quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
from tools.telemetry import ClockRegion, Sampler

dev = torch.device("cuda", 0)
quick = "--quick" in sys.argv
rows = []


def region(name, fn, flop=0.0, byts=0.0, target_s=0.6):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    n = max(20, int((0.25 if quick else target_s) / (e0.elapsed_time(e1) / 10 / 1e3)))
    r = ClockRegion(dev)
    with Sampler(0, 0.004) as sm:
        e0.record(); r.begin()
        for _ in range(n):
            fn()
        r.end(); e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    s = sm.summary()
    # drop the ramp: average of the samples of the second half of the region
    pw = [x[0] / 1e6 for x in sm.rows[len(sm.rows) // 2:] if x[0] is not None]
    w = sum(pw) / max(1, len(pw))
    rows.append((name, us, r.mhz() or 0.0, w, w * us * 1e-6, flop / us / 1e6 if flop else 0.0, byts / us / 1e6 if byts else 0.0))
    print(f"{name:46s} {us:9.1f} us  sclk {r.mhz() or 0:6.0f} MHz  {w:6.0f} W  {w * us * 1e-3:8.2f} mJ/launch"
          + (f"  {flop / us / 1e6:7.1f} TF/s  {w * us * 1e-6 / flop * 1e12:6.2f} pJ/flop" if flop else "")
          + (f"  {byts / us / 1e6:5.2f} TB/s  {w * us * 1e-6 / byts * 1e12:6.1f} pJ/B" if byts else ""), flush=True)
    time.sleep(0.3)


def bf(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


# ---- GEMMs (the step's shapes) ------------------------------------------------------------------------------------------------
def gemm_case(name, M, N, K, tA=False, tB=False, opts=None, **kw):
    A = bf(K, M) if tA else bf(M, K)
    Bm = bf(K, N) if tB else bf(N, K)
    f32 = tA
    Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    ws = torch.empty(8 * 3072 * 768, device=dev) if tA else None
    extra = {}
    if kw.get("z"):
        extra = dict(dact=L.ACT_RELU, z=bf(M, N).relu_(), dropout_p=0.1, dropout_seed=3)
    if kw.get("res"):
        extra = dict(residual=bf(M, N), dropout_p=0.1, dropout_seed=3)
    if kw.get("relu"):
        extra = dict(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3)
    old = {k: L.get_option(k) for k in (opts or {})}
    for k, v in (opts or {}).items():
        L.set_option(k, v)
    region(name, lambda: L.gemm(A, Bm, Cm, M, N, K, transA=tA, transB=tB, workspace=ws, **extra), flop=2.0 * M * N * K)
    for k, v in old.items():
        L.set_option(k, v)


gemm_case("enc QKV fwd 32000x2304x768 (a4p)", 32000, 2304, 768)
gemm_case("enc QKV fwd 32000x2304x768 (gemm_a4=0: p8)", 32000, 2304, 768, opts={"gemm_a4": 0})
gemm_case("enc wi fwd relu+drop 32000x3072x768 (a4p)", 32000, 3072, 768, relu=True)
gemm_case("enc wo fwd res+drop 32000x768x3072 (dma)", 32000, 768, 3072, res=True)
gemm_case("enc o fwd res+drop 32000x768x768 (dma)", 32000, 768, 768, res=True)
gemm_case("enc wo dgrad masked 32000x3072x768 (dma)", 32000, 3072, 768, tB=True, z=True)
gemm_case("enc wo dgrad masked 32000x3072x768 (a4p, =5)", 32000, 3072, 768, tB=True, z=True, opts={"gemm_a4": 5})
gemm_case("enc wi dgrad 32000x768x3072 (a4p)", 32000, 768, 3072, tB=True)
gemm_case("enc qkv dgrad 32000x768x2304 (a4p)", 32000, 768, 2304, tB=True)
gemm_case("enc wi wgrad 3072x768x32000 (p8 + splitk)", 3072, 768, 32000, tA=True, tB=True)
gemm_case("enc wi wgrad 3072x768x32000 (a4, =4)", 3072, 768, 32000, tA=True, tB=True, opts={"gemm_a4": 4})
gemm_case("enc qkv wgrad 2304x768x32000 (p8 + splitk)", 2304, 768, 32000, tA=True, tB=True)
gemm_case("dec QKV fwd 8192x2304x768", 8192, 2304, 768)
gemm_case("dec wo fwd res+drop 8192x768x3072", 8192, 768, 3072, res=True)
gemm_case("dec wi dgrad 8192x768x3072", 8192, 768, 3072, tB=True)
gemm_case("vit fc2 fwd 3200x768x2048", 3200, 768, 2048)


# ---- attention (encoder layer shape) ------------------------------------------------------------------------------------------
B, H, N = 32, 12, 1000
W = H * 64
qkv = bf(B, N, 3 * W); d_o = bf(B, N, W)
o = torch.empty(B, N, W, dtype=torch.bfloat16, device=dev); ml = torch.empty(B, H, N, 2, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, 4, device=dev)
diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
lens = torch.randint(700, 1001, (B,), device=dev)
mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
st = (N * 3 * W, 3 * W)
a = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o, st, st, st, (N * W, W), ml=ml, scale=1.0, bias_diag=diag, key_mask=mask,
                dropout_p=0.1, dropout_seed=5)
fl = 4.0 * B * H * N * N * 64
region("enc attn fwd (bias, mask, dropout)", lambda: L.attn_fwd(a), flop=fl)
L.attn_fwd(a)
for part, nm, f in ((1, "enc attn bwd dQ", fl * 1.5), (2, "enc attn bwd dK/dV", fl * 2.0), (0, "enc attn bwd both", fl * 2.0)):
    L.set_option("attn_bwd_part", part)
    region(nm, lambda: L.attn_bwd(a, d_o, (N * W, W), delta, dqkv, dqkv[..., W:], dqkv[..., 2 * W:], st, st, st, dbias_diag=ddiag, far=(-91, 91)), flop=f)
L.set_option("attn_bwd_part", 0)

# ---- HBM-bound kernels --------------------------------------------------------------------------------------------------------
M, d = 32000, 768
x = bf(M, d); w = torch.ones(d, device=dev); rstd = torch.rand(M, device=dev) + 0.5
dy = bf(M, d); dx = torch.empty_like(x); dadd = bf(M, d); dw = torch.zeros(d, device=dev); dxd = torch.empty_like(x); y = torch.empty_like(x)
region("rmsnorm fwd 32000x768", lambda: L.rmsnorm_fwd(x, w, y, rstd, M, d, 1e-6), byts=M * d * 2 * 2)
region("rmsnorm bwd (+res, +dropout out) 32000x768", lambda: L.rmsnorm_bwd(x, w, rstd, dy, dx, dadd, dw, M, d, dx_drop=dxd, dropout_p=0.1, dropout_seed=3), byts=M * d * 2 * 5)
src = torch.randn(64 << 20, device=dev); dst = torch.empty_like(src)
region("HBM copy 256 MB (torch)", lambda: dst.copy_(src), byts=2 * 4 * (64 << 20))

print()
print("| kernel | us / launch | effective sclk MHz | socket W | mJ / launch | TF/s | TB/s |")
print("|---|---|---|---|---|---|---|")
for n, us, mhz, w_, j, tf, tb in rows:
    print(f"| {n} | {us:.1f} | {mhz:.0f} | {w_:.0f} | {j * 1e3:.2f} | {tf:.0f} | {tb:.2f} |")
