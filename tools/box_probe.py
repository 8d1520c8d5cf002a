"""What can an ordinary user read about clock and power on the GPU box?    Lists the hwmon / sysfs files of the GPU,
tries rocm-smi and amdsmi, and checks v2s_clock_probe against a known load: idle, an HBM copy loop, a K = 768 GEMM loop, an 8192^3 GEMM loop
(effective shader clock from s_memtime / s_memrealtime vs HIP-event time)."""
import glob, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

print("== sysfs")
for f in sorted(glob.glob("/sys/class/drm/card0/device/hwmon/hwmon*/*")) + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")) + \
        sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent")) + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk")):
    if os.path.isfile(f):
        try:
            v = open(f).read().strip().replace("\n", " | ")[:100]
        except Exception as e:
            v = f"<{type(e).__name__}>"
        print(f, "=", v)
for cmd in (["rocm-smi", "--showpower", "--showclocks", "--json"], ["amd-smi", "metric", "-p", "-c", "--json"]):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        print("==", " ".join(cmd), "rc", r.returncode); print(r.stdout[:1500]); print(r.stderr[:300])
    except Exception as e:
        print("==", " ".join(cmd), "failed:", e)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    print("== amdsmi power:", amdsmi.amdsmi_get_power_info(h))
    print("== amdsmi clock:", amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
except Exception as e:
    print("== amdsmi python failed:", type(e).__name__, e)

dev = torch.device("cuda", 0)


from tools.telemetry import hwmon_dir
HW = hwmon_dir(0)
print("== hwmon of torch device 0:", HW)


def power_files():
    return [HW + "/power1_input"] if HW else []


def region(name, fn, n=2000, sleep=0.0):
    """n back-to-back launches (no host sync in between: the GPU never idles inside the region), clock probes around them, power / driver
    clock sampled by a host thread every 5 ms"""
    from tools.telemetry import ClockRegion, Sampler
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    r = ClockRegion(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with Sampler(0, 0.005) as sm:
        e0.record(); r.begin()
        for _ in range(n):
            fn()
        r.end(); e1.record()
        if sleep:
            time.sleep(sleep)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{name:34s} {ms:8.1f} ms ({ms / n * 1e3:8.1f} us/iter)  effective sclk {r.mhz() or 0:7.1f} MHz   {sm.summary()}")


print("== clock probe vs load")
x = torch.randn(64 << 20, device=dev); y = torch.empty_like(x)
A = torch.randn(32000, 768, device=dev).bfloat16(); W = torch.randn(2304, 768, device=dev).bfloat16(); Cc = torch.empty(32000, 2304, device=dev, dtype=torch.bfloat16)
A8 = torch.randn(8192, 8192, device=dev).bfloat16(); B8 = torch.randn(8192, 8192, device=dev).bfloat16(); C8 = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
region("idle (host sleeps 1 s)", lambda: None, n=1, sleep=1.0)
region("HBM copy 256 MB", lambda: y.copy_(x), n=6000)
region("gemm 32000x2304x768 (a4p)", lambda: L.gemm(A, W, Cc, 32000, 2304, 768), n=10000)
L.set_option("gemm_a4", 0)
region("gemm 32000x2304x768 (gemm_a4=0)", lambda: L.gemm(A, W, Cc, 32000, 2304, 768), n=10000)
L.set_option("gemm_a4", 1)
region("gemm 8192^3", lambda: L.gemm(A8, B8, C8, 8192, 8192, 8192), n=1500)
Z = torch.zeros_like(A8)
region("gemm 8192^3 zeros", lambda: L.gemm(Z, Z, C8, 8192, 8192, 8192), n=1500)
region("idle again", lambda: None, n=1, sleep=1.0)
