"""What state the chip was in while a section ran: the transforms are VALU-bound on a power-limited part,
so a fraction of the HBM roofline means little without the shader clock beside it.  A background thread reads the amdgpu hwmon
files of the GPU this rank uses (freq1_input = sclk in Hz, power1_input = socket power in microwatts) every few milliseconds;
`with sampler.section("name")` records mean / min / max over the section.  No file, no numbers (the fields are then null)."""
from __future__ import annotations

import glob
import os
import threading
import time


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _hwmon_dir(pci_bus_id: str | None):
    """hwmon directory of the amdgpu device with this PCI address (e.g. '0000:05:00.0'); the only one with an sclk label otherwise"""
    cands = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        try:
            if open(os.path.join(d, "freq1_label")).read().strip() != "sclk":
                continue
        except OSError:
            continue
        cands.append(d)
    if pci_bus_id:
        for d in cands:
            if os.path.realpath(os.path.join(d, "..", "..")).lower().endswith(pci_bus_id.lower()):
                return d
    return cands[0] if len(cands) == 1 else None


class ChipSampler:
    def __init__(self, torch=None, device_index: int = 0, period_s: float = 0.004):
        bus = None
        try:
            p = torch.cuda.get_device_properties(device_index)
            bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:
            pass
        self.dir = None if os.environ.get("HP_BENCH_NO_CHIP") else _hwmon_dir(bus)   # (HP_BENCH_NO_CHIP: A/B of the sampler's own cost)
        self.period = period_s
        self.samples = []            # (t, sclk_MHz, power_W)
        self.sections = {}
        self._stop = threading.Event()
        self._thread = None
        if self.dir:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def _read(self, name, scale):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read()) / scale
        except (OSError, ValueError):
            return None

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(), self._read("freq1_input", 1e6), self._read("power1_input", 1e6)))
            time.sleep(self.period)

    def close(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)

    def section(self, name):
        return _Section(self, name)

    def summarize(self, t0, t1):
        rows = [s for s in self.samples if t0 <= s[0] <= t1]
        clk = [s[1] for s in rows if s[1] is not None]
        pw = [s[2] for s in rows if s[2] is not None]
        if not clk:
            return {"sclk_MHz": None, "socket_power_W": None, "samples": 0}
        return {"sclk_MHz": round(sum(clk) / len(clk), 1), "sclk_MHz_min": min(clk), "sclk_MHz_max": max(clk),
                "socket_power_W": round(sum(pw) / len(pw), 1) if pw else None, "samples": len(clk), "seconds": round(t1 - t0, 3)}


class _Section:
    def __init__(self, sampler, name):
        self.s, self.name = sampler, name

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.s.sections[self.name] = self.s.summarize(self.t0, time.perf_counter())
        return False
