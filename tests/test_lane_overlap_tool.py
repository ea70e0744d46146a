"""tools/lane_overlap.py: the interval arithmetic behind `kernels in flight` and CU-slot-time (no GPU: hand-made stamps)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lane_overlap as lo                                                   # noqa: E402


def test_two_lanes_half_overlapped():
    """Two lanes, one launch per step of 100 us (10 000 ticks), lane 1 offset by 50 us: outside the ends two kernels overlap half of
    the time -> 1.0 kernels while busy would be no overlap, 2.0 full overlap; here every instant of the steady state has exactly 2."""
    iv = []
    for step in range(40):
        lane = step % 2
        t0 = 1_000_000 + step * 5_000                                       # a new kernel every 50 us, each 100 us long
        iv.append((lane, step, 0, t0, t0 + 10_000))
    launches = [dict(kernel="k", workgroups=128, threads=256, lds_bytes=0, wg_per_cu=2, registers=64)]
    r = lo.analyse(iv, launches, fps=1.0, ms_step=0.05, lanes=2, steps=40, batch=8, fps_product=None)
    assert abs(r["kernels_in_flight_mean_while_busy"] - 2.0) < 0.06 and r["chip_idle_share"] == 0.0
    assert abs(r["us_per_step_device_clock"] - 50.0) < 2.0
    p = r["per_launch"][0]
    assert p["mean_us"] == 100.0 and p["chip_share"] == 0.25 and p["slot_time_us"] == 25.0       # 128 of 512 workgroup slots for 100 us
    assert r["cu_slot_time_us_per_step"] == 25.0 and abs(r["slot_time_over_step_time"] - 0.5) < 0.03
    assert "KERNELS IN FLIGHT: mean 2.0" in lo.render(r, "test")


def test_serial_lanes_and_bad_stamps():
    iv = [(s % 4, s, 0, 1000 + s * 1000, 1000 + s * 1000 + 400) for s in range(40)]               # one at a time, chip idle 60 %
    iv.append((0, 20, 0, 0, 5))                                                                   # a pair that was never written: dropped
    launches = [dict(kernel="k", workgroups=4096, threads=256, lds_bytes=0, wg_per_cu=4, registers=64)]
    r = lo.analyse(iv, launches, 1.0, 0.01, lanes=4, steps=40, batch=8, fps_product=123.0)
    assert r["intervals_dropped"] == 1 and r["kernels_in_flight_mean_while_busy"] == 1.0
    assert 0.55 < r["chip_idle_share"] < 0.62
    assert r["per_launch"][0]["chip_share"] == 1.0 and r["per_launch"][0]["rounds"] == 4.0       # more workgroups than slots: the whole chip, four rounds
