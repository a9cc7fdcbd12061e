"""tools/pmc_live.py (bench.py's live roofline.traffic): the counter CSV -> bytes arithmetic on a synthetic rocprofv3 counter_collection
file, the kernel-name patterns, and the profiler-detection guard.  No GPU, no rocprofv3: the two passes are replaced by fixtures."""
import csv
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    spec = importlib.util.spec_from_file_location("pmc_live", os.path.join(ROOT, "tools", "pmc_live.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SX = "void chip::bal_f9_s0::(anonymous namespace)::bal_stream_kernel<0, true, true, false>(chip::BalArgs)"
JT = "void chip::bal_f9_s0::(anonymous namespace)::bal_stream_kernel<1, true, true, false>(chip::BalArgs)"
RED = "chip::bal_f9_s0::(anonymous namespace)::bal_reduce_partials_kernel(double const*, int, int, chip::FMap, double const*)"
SPILL = "void chip::bal_f9_s0::(anonymous namespace)::bal_stream_kernel<0, false, true, false>(chip::BalArgs)"
CHUNK = "chip::bal_f9_s0::(anonymous namespace)::bal_camera_chunk_kernel(chip::ZUnits, double const*, double*)"


def write_csv(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (k, v) in enumerate(rows):
            w.writerow({"Dispatch_Id": i, "Kernel_Name": k, "Counter_Name": counter, "Counter_Value": v})
            w.writerow({"Dispatch_Id": i, "Kernel_Name": k, "Counter_Name": "SOMETHING_ELSE", "Counter_Value": 1e9})


def test_counter_means(tmp_path):
    m = load()
    p = tmp_path / "c.csv"
    write_csv(p, "FETCH_SIZE", [(SX, 100.0), (SX, 300.0), (RED, 10.0)])
    means, counts = m._counter_means(str(p), "FETCH_SIZE")
    assert means == {SX: 200.0, RED: 10.0} and counts == {SX: 2, RED: 1}


@pytest.mark.parametrize("spill", [False, True])
def test_measure_sums_like_make_pmc_traffic(monkeypatch, spill):
    """bytes = 1024 (2 FETCH_SIZE + WRITE_SIZE) over the tile kernel + the reduction (+ the camera-major pass when the cameras spill)."""
    m = load()
    tile = SPILL if spill else SX
    fetch = {tile: 500000.0, RED: 20000.0, JT: 510000.0, CHUNK: 7000.0}
    write = {tile: 30000.0, RED: 100.0, JT: 55000.0, CHUNK: 300.0}
    if not spill:
        fetch.pop(CHUNK), write.pop(CHUNK)

    def fake(counter, workload, timeout, env_extra):
        t = fetch if counter == "FETCH_SIZE" else write
        return dict(t), {k: 15 for k in t}
    monkeypatch.setattr(m, "_one_pass", fake)
    out = m.measure("venice1778")
    want_sx = 1024 * (2 * (500000 + 20000 + (7000 if spill else 0)) + 30000 + 100 + (300 if spill else 0))
    assert out["sx"] == want_sx
    assert out["jtjx"] == 1024 * (2 * (510000 + 20000) + 55000 + 100)
    assert set(out["breakdown_KiB"]["sx"]) == ({"bal_stream_kernel<0, false(, (true|false)){0,2}>", "bal_reduce_partials_kernel", "bal_camera_chunk_kernel"} if spill
                                                else {"bal_stream_kernel<0, true(, (true|false)){0,2}>", "bal_reduce_partials_kernel"})


def test_measure_raises_without_operator_kernels(monkeypatch):
    m = load()
    monkeypatch.setattr(m, "_one_pass", lambda *a: ({"some_other_kernel": 1.0}, {"some_other_kernel": 1}))
    with pytest.raises(RuntimeError):
        m.measure("venice1778")


def test_profiler_guard(monkeypatch):
    m = load()
    for k in list(os.environ):
        if k.startswith(("ROCPROF_", "ROCPROFILER_", "ROCP_TOOL")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not m.under_a_profiler()
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert m.under_a_profiler()
    monkeypatch.delenv("ROCPROFILER_LIBRARY_CTOR")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert m.under_a_profiler()
