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
