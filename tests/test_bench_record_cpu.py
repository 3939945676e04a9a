"""Host-side pieces of bench.py that need no GPU: the compact per-configuration summary the record carries inside
`roofline` (so that a parser which keeps only the contract's keys still sees every configuration), and the device check
of the multi-GPU record."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_configs_summary_keeps_what_a_reader_needs():
    rows = [
        {"config": "cfg1 2D multilinear::regular 4x4, 1e3 obs", "oracle_check": {"bitwise_equal": True}},
        {"config": "cfg2 3D multilinear::regular 64^3, 1e8 obs", "kernel_ms": 0.91, "frac": 0.44, "oracle_check": {"bitwise_equal": True},
         "traffic": {"ratio_to_algorithmic": 1.97, "fabric_requests_per_point": 0.55}},
        {"config": "cfg4 4D multicubic::regular 32^4, 1e7 obs, linearize_extrapolation=true", "kernel_ms": 0.83, "frac": 0.06,
         "oracle_check": {"bitwise_equal": True}, "binned": {"stage_ms": {"hist": 0.04, "scan": 0.01, "scatter": 0.19, "kernel": 0.6}},
         "traffic": {"ratio_to_algorithmic": 6.3, "fabric_requests_per_point": 3.17}},
        {"error": "RuntimeError('x')"},
    ]
    s = bench.configs_summary(rows)
    assert [e.get("cfg") for e in s[:3]] == ["cfg1", "cfg2", "cfg4 lin"]
    assert s[1] == {"cfg": "cfg2", "kernel_ms": 0.91, "frac": 0.44, "bitwise_equal": True, "traffic_ratio": 1.97, "fabric_requests_per_point": 0.55}
    assert s[2]["stage_ms"]["kernel"] == 0.6 and s[2]["bitwise_equal"] is True
    assert s[0]["kernel_ms"] is None and s[3] == {"error": "RuntimeError('x')"}


def test_distinct_device_check():
    devs = [{"rank": r, "cuda_device": r, "pci": "0000:%02x:00" % (0x10 + r)} for r in range(8)]
    bench.check_distinct_devices(devs, 8)  # eight ranks on eight GPUs
    shared = [dict(d, pci="0000:07:00") for d in devs]
    with pytest.raises(SystemExit, match="distinct device"):
        bench.check_distinct_devices(shared, 8)
    # no PCI address from the runtime: the device indices decide
    nopci = [dict(d, pci=None) for d in devs]
    bench.check_distinct_devices(nopci, 8)
    with pytest.raises(SystemExit):
        bench.check_distinct_devices([dict(d, pci=None, cuda_device=0) for d in devs], 8)
