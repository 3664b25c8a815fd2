"""CPU: measurement helpers of bench.py that do not need a GPU (tools/gpu_power.py)."""
import os

from tools import gpu_power


def test_pick_hwmon_selects_the_card_under_load(tmp_path):
    dirs = {}
    for name, (idle, busy) in {"card0": (240e6, 241e6), "card24": (290e6, 1400e6), "card8": (1260e6, 1290e6)}.items():
        d = tmp_path / name / "device" / "hwmon" / "hwmon1"
        d.mkdir(parents=True)
        (d / "freq1_input").write_text("1708000000\n")
        (d / "power1_input").write_text(str(int(busy)) + "\n")
        (d / "power1_cap").write_text("1400000000\n")
        dirs[str(d)] = (int(idle), int(busy))
    files, info = gpu_power.pick_hwmon({d: v[0] for d, v in dirs.items()}, {d: v[1] for d, v in dirs.items()})
    assert "card24" in files["sclk_hz"] and info["power_cap_w"] == 1400.0 and info["idle_power_w"] == 290.0
    s = gpu_power.Sampler(files, period=0.001)
    s.start()
    import time
    time.sleep(0.05)
    s.stop()
    sm = s.summary()
    assert sm["sclk_mhz"]["avg"] == 1708.0 and sm["power_w"]["max"] == 1400.0 and sm["samples"] > 3
    # nothing rose by 100 W: no pick
    assert gpu_power.pick_hwmon({d: v[0] for d, v in dirs.items()}, {d: v[0] + 1 for d, v in dirs.items()}) == ({}, None)
    assert gpu_power.pick_hwmon({}, {}) == ({}, None)


def test_pick_hwmon_prefers_the_pci_address(tmp_path):
    """two GPUs under load (another tenant's draws more): the PCI address of our device decides."""
    snaps = {}
    for name, pci, (idle, busy) in (("card8", "0000:0a:00.0", (250e6, 1400e6)), ("card48", "0000:d9:00.0", (246e6, 1300e6))):
        dev = tmp_path / "pci" / pci
        hw = dev / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (hw / "freq1_input").write_text("2000000000\n")
        (hw / "power1_input").write_text(str(int(busy)) + "\n")
        card = tmp_path / "drm" / name
        card.mkdir(parents=True)
        os.symlink(dev, card / "device")
        snaps[str(card / "device" / "hwmon" / "hwmon3")] = (int(idle), int(busy))
    idle = {d: v[0] for d, v in snaps.items()}
    busy = {d: v[1] for d, v in snaps.items()}
    files, info = gpu_power.pick_hwmon(idle, busy, pci="0000:d9:00.0")
    assert "card48" in files["power_uw"] and info["picked_by"] == "pci_address"
    files, info = gpu_power.pick_hwmon(idle, busy)                       # no address: the larger rise
    assert "card8" in files["power_uw"] and info["picked_by"] == "power_rise"
    files, info = gpu_power.pick_hwmon(idle, busy, pci="0000:ff:00.0")   # unknown address: falls back
    assert "card8" in files["power_uw"] and info["picked_by"] == "power_rise"
