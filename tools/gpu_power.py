"""Shader clock / board power of the GPU this process computes on, from the amdgpu hwmon files (measurement helper of
bench.py and tools/clock_probe.py; host-side file reads only, nothing on the GPU)."""
import glob
import os
import threading
import time


def read_int(path):
    try:
        return int(open(path).read())
    except (OSError, ValueError):
        return None


def _powers():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        v = read_int(os.path.join(d, "power1_input"))
        if v is None:
            v = read_int(os.path.join(d, "power1_average"))
        if v is not None:
            out[d] = v
    return out


def snapshot():
    """Board power (uW) of every amdgpu hwmon directory: take one idle and one under load, then pick_hwmon()."""
    return _powers()


def pci_address(device_index=0):
    """'dddd:bb:dd.0' of a torch device (None when torch / the properties are not available)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        return None


def pick_hwmon(idle, busy, min_rise_w=100.0, pci=None):
    """The hwmon directory of this process's GPU (a node lists every GPU and partition under /sys/class/drm): the one
    whose PCI address is `pci` if its power rose between the two snapshots, otherwise the one whose power rose most
    (other tenants' GPUs may be busy too: the PCI match is the safer key).  Returns (files, info) or ({}, None)."""
    common = [d for d in idle if d in busy]
    if not common:
        return {}, None
    d, how = None, "power_rise"
    if pci:
        for c in common:
            try:
                dev = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(c))))
            except OSError:
                continue
            if dev == pci and busy[c] - idle[c] >= min_rise_w * 1e6:
                d, how = c, "pci_address"
                break
    if d is None:
        d = max(common, key=lambda k: busy[k] - idle[k])
    if busy[d] - idle[d] < min_rise_w * 1e6:
        return {}, None
    files = {}
    for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_input", "power1_average"))):
        for n in names:
            if key not in files and read_int(os.path.join(d, n)) is not None:
                files[key] = os.path.join(d, n)
    info = {"card": next((c for c in d.split("/") if c.startswith("card")), d), "picked_by": how,
            "idle_power_w": idle[d] / 1e6,
            "power_cap_w": (read_int(os.path.join(d, "power1_cap")) or 0) / 1e6}
    try:
        info["sclk_levels"] = " ".join(open(os.path.join(os.path.dirname(os.path.dirname(d)), "pp_dpm_sclk")).read().split())
    except OSError:
        pass
    return files, info


class Sampler(threading.Thread):
    """Reads the files every `period` seconds until stop(); summary() -> averages in MHz / W."""

    def __init__(self, files, period=0.01):
        super().__init__(daemon=True)
        self.files, self.period, self.rows, self._stop_flag = files, period, [], False

    def run(self):
        while not self._stop_flag:
            self.rows.append({k: read_int(p) for k, p in self.files.items()})
            time.sleep(self.period)

    def stop(self):
        self._stop_flag = True
        self.join()

    def summary(self, drop_first=0.25):
        rows = self.rows[int(len(self.rows) * drop_first):]
        out = {"samples": len(rows)}
        for k, name, div in (("sclk_hz", "sclk_mhz", 1e6), ("power_uw", "power_w", 1e6)):
            v = [r[k] for r in rows if r.get(k) is not None]
            if v:
                out[name] = {"avg": round(sum(v) / len(v) / div, 1), "min": round(min(v) / div, 1), "max": round(max(v) / div, 1)}
        return out
