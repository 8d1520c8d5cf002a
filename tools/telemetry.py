"""Clock / power telemetry an ordinary user can read on the GPU box (measurement support for bench.py and the A/B tools; not product code).

* effective shader clock of a region: two v2s_clock_probe stamps (s_memtime cycles / s_memrealtime 100 MHz ticks per XCD) -> MHz.  The
  cycle counter stands still while the XCD idles, so this is cycles actually clocked per wall second: idle gaps pull it down, power
  throttling pulls it down (a K = 768 GEMM loop alone runs at ~1.34 GHz of the 2.4 GHz maximum, profiles/r06_box_probe.txt).
* socket power / junction temperature / the driver's own sclk reading of THIS process's GPU: the amdgpu hwmon files of the PCI device torch
  runs on (the host shows all eight cards), sampled by a host thread."""
import glob
import os
import threading
import time

import torch


def hwmon_dir(dev=0):
    """/sys/class/drm/cardN/device/hwmon/hwmonM of the torch device (matched by PCI address), or None"""
    try:
        p = torch.cuda.get_device_properties(dev)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except Exception:
        return None
    for card in glob.glob("/sys/class/drm/card*/device"):
        try:
            if os.path.basename(os.path.realpath(card)).lower().startswith(want):
                hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                if hw:
                    return hw[0]
        except OSError:
            pass
    return None


def _read(path):
    try:
        return int(open(path).read())
    except (OSError, ValueError):
        return None


class Sampler:
    """with Sampler(dev) as s: ...   ->  s.summary() = {"power_w_avg", "power_w_max", "sclk_mhz_driver_avg", "temp_c_max", "samples"}"""
    def __init__(self, dev=0, period_s=0.02):
        self.dir, self.period, self.rows, self._stop = hwmon_dir(dev), period_s, [], threading.Event()
        self._t = None

    def _run(self):
        d = self.dir
        while not self._stop.is_set():
            self.rows.append((_read(d + "/power1_input"), _read(d + "/freq1_input"), _read(d + "/temp2_input")))
            time.sleep(self.period)

    def __enter__(self):
        if self.dir is not None:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t is not None:
            self._t.join()

    def summary(self):
        pw = [r[0] / 1e6 for r in self.rows if r[0] is not None]
        fq = [r[1] / 1e6 for r in self.rows if r[1] is not None]
        tp = [r[2] / 1e3 for r in self.rows if r[2] is not None]
        if not self.rows:
            return {"samples": 0, "note": "no readable amdgpu hwmon files for this device"}
        out = {"samples": len(self.rows), "source": self.dir}
        if pw:
            out.update(power_w_avg=round(sum(pw) / len(pw), 1), power_w_max=round(max(pw), 1))
            cap = _read(self.dir + "/power1_cap")
            if cap:
                out["power_cap_w"] = round(cap / 1e6, 1)
        if fq:
            out["sclk_mhz_driver_avg"] = round(sum(fq) / len(fq), 1)
        if tp:
            out["temp_junction_c_max"] = round(max(tp), 1)
        return out


class ClockRegion:
    """r = ClockRegion(dev); r.begin(); ...launches...; r.end(); torch.cuda.synchronize(); r.mhz()"""
    def __init__(self, dev):
        from vidchapters_amd import lib as L
        self.L = L
        self.p0 = torch.zeros(8, 4, dtype=torch.int64, device=dev)
        self.p1 = torch.zeros(8, 4, dtype=torch.int64, device=dev)

    def begin(self):
        self.L.clock_probe(self.p0)

    def end(self):
        self.L.clock_probe(self.p1)

    def mhz(self):
        v = self.L.effective_sclk_mhz(self.p0.cpu(), self.p1.cpu())
        return round(v, 1) if v else None
