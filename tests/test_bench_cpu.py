"""Host-side pieces of bench.py that the N > 1 line depends on (no GPU): the stall watchdog and the a-priori link arithmetic."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_link_bound_estimate_arithmetic():
    import bench
    assert bench.link_bound_estimate(2_620_000_000, 1, 5.9) is None
    e = bench.link_bound_estimate(2_620_000_000, 8, 5.9)
    # 2 (N - 1) / N of the gradient bytes cross every rank's links: 4.585 GB at N = 8
    assert e['bytes_per_rank_on_the_wire'] == int(2.0 * 7 / 8 * 2_620_000_000)
    assert e['all_links_ms'] == pytest.approx(1e3 * e['bytes_per_rank_on_the_wire'] / (7 * 153e9))
    assert e['single_ring_ms'] == pytest.approx(3.5 * e['all_links_ms'])
    assert 0.0 < e['weak_scaling_ceiling_if_fully_exposed'] < 1.0
    e2 = bench.link_bound_estimate(2_620_000_000, 2, 5.9)
    assert e2['all_links_ms'] == pytest.approx(1e3 * 2_620_000_000 / 153e9)      # one peer: one link


def test_watchdog_ends_a_stalled_rank_with_code_3():
    """A rank that stops beating exits by itself (code 3) -- the launcher (and torch.distributed.run) then tear the job down;
    a rank that keeps beating is left alone."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Watchdog(1.5, 0)\n"
            "for _ in range(8):\n"
            "    time.sleep(0.3); d.beat('loop')\n"
            "print('alive', flush=True)\n"
            "time.sleep(30)\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert 'alive' in r.stdout and 'no progress' in r.stderr and 'loop' in r.stderr
