"""Box calibration and clock / power sampling for bench.py and the A/B scripts (round 6).

The MI355X runs the FlowDec step at its POWER limit (2.10 GHz at 1300 W of a 2.4 GHz part, profiles/r05_power_probe.txt) and the
boxes of a pool differ by +-2.5 % on MFMA-heavy kernels, so a wall-clock number alone cannot tell a code change from another box, nor
an energy saving from a stall shuffle.  This module gives every measurement two companions:

* `calibrate(device)`: a fixed matrix-core issue loop (fd_calibrate_mfma: register-resident random bf16, no memory) -> the TFLOP/s
  the box sustains, with the shader clock and package power sampled WHILE it runs;
* `PowerSampler`: a background thread that samples shader clock and package power of one GPU (amdgpu hwmon files in sysfs; the
  `rocm-smi` text output as a fallback) around any region: `with PowerSampler(dev) as ps: ...; ps.summary()`.

Host-side measurement infrastructure only: nothing here is on the enhance() path, and the reference has no counterpart (its driver
times one call with two CUDA events, enhance.py:120-136)."""
import ctypes as C
import glob
import os
import re
import subprocess
import threading
import time


def _pci_bus_id(device) -> str:
    import torch
    p = torch.cuda.get_device_properties(device)
    # torch exposes domain / bus / device numbers; sysfs names the function 0000:bb:dd.0
    return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))


def find_hwmon(device=0):
    """-> the amdgpu hwmon directory of `device` (matched by PCI address; the only amdgpu hwmon if exactly one is visible), or None."""
    cands = []
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        try:
            with open(os.path.join(hw, "name")) as f:
                if f.read().strip() != "amdgpu":
                    continue
        except OSError:
            continue
        cands.append((os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))), hw))
    if not cands:
        return None
    try:
        want = _pci_bus_id(device)
        for pci, hw in cands:
            if pci.lower() == want.lower():
                return hw
    except Exception:
        pass
    uniq = sorted({hw for _, hw in cands})
    return uniq[0] if len(uniq) == 1 else None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def read_hwmon(hw):
    """-> (sclk MHz or None, package power W or None) from one amdgpu hwmon directory."""
    sclk = _read_int(os.path.join(hw, "freq1_input"))                     # Hz
    pw = _read_int(os.path.join(hw, "power1_average"))                    # microwatts (older / newer firmware expose one of the two)
    if pw is None:
        pw = _read_int(os.path.join(hw, "power1_input"))
    return (sclk / 1e6 if sclk else None, pw / 1e6 if pw else None)


_SMI_SCLK = re.compile(r"sclk.*?\((\d+)\s*Mhz\)", re.I)
_SMI_POWER = re.compile(r"(?:Package|Socket)[^:\n]*Power[^:\n]*:\s*([0-9.]+)", re.I)


def read_rocm_smi():
    """Fallback: one `rocm-smi --showpower --showclocks` call (~0.3 s), first GPU listed.  -> (sclk MHz, W), None where not found."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except (OSError, subprocess.SubprocessError):
        return None, None
    a, b = _SMI_SCLK.search(out), _SMI_POWER.search(out)
    return (float(a.group(1)) if a else None, float(b.group(1)) if b else None)


class PowerSampler:
    """Samples (shader clock, package power) of one GPU in a background thread while the `with` body runs."""

    def __init__(self, device=0, period_s=0.05):
        self.hw = find_hwmon(device)
        self.period = period_s if self.hw else max(period_s, 0.5)      # the rocm-smi fallback is a process per sample
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    def _loop(self):
        while not self._stop.is_set():
            s = read_hwmon(self.hw) if self.hw else read_rocm_smi()
            if s[0] is not None or s[1] is not None:
                self.samples.append(s)
            self._stop.wait(self.period)

    def __enter__(self):
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=15)
        return False

    def summary(self):
        """Mean / max over the samples (None when the box exposes neither source)."""
        clk = [s[0] for s in self.samples if s[0] is not None]
        pw = [s[1] for s in self.samples if s[1] is not None]
        return {"sclk_mhz": sum(clk) / len(clk) if clk else None, "sclk_mhz_max": max(clk) if clk else None,
                "power_w": sum(pw) / len(pw) if pw else None, "power_w_max": max(pw) if pw else None,
                "samples": len(self.samples), "source": "hwmon" if self.hw else "rocm-smi"}


def calibrate(device=0, repeats=5, iters=0):
    """`box_calibration` of a bench line: the fixed MFMA issue loop for ~0.25 s with clock / power sampled while it runs."""
    import torch
    from . import _lib as L
    lib = L.load()
    dev = torch.device("cuda", device) if not isinstance(device, torch.device) else device
    with torch.cuda.device(dev):
        scratch = torch.zeros(512, dtype=torch.float32, device=dev)
        tf, ms = C.c_double(), C.c_double()
        torch.cuda.synchronize(dev)
        with PowerSampler(dev.index or 0, period_s=0.02) as ps:
            L.check(lib.fd_calibrate_mfma(L.ptr(scratch), int(iters), int(repeats), C.byref(tf), C.byref(ms), L.stream()))
        s = ps.summary()
    return {"mfma_tflops": tf.value, "ms": ms.value, "sclk_mhz": s["sclk_mhz"], "power_w": s["power_w"], "samples": s["samples"],
            "source": s["source"], "kernel": "fd_calibrate_mfma: v_mfma_f32_32x32x16_bf16, register-resident random bf16, 2 x 8 waves per CU, "
                                             f"{repeats} launches"}


if __name__ == "__main__":
    import json
    print(json.dumps(calibrate()))
    time.sleep(0)
