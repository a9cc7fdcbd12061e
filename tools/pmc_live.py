#!/usr/bin/env python3
"""HBM traffic of the dominant operators measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE — the TCC has four
counter slots, FETCH_SIZE takes three; /opt/skills/guides/MI355X_MICROARCH.md's HBM section) around a child that applies S.x and
JtJx on the workload (tools/kernel_times.py <workload> --operators-only), summed per operator exactly as tools/make_pmc_traffic.py
sums the committed summaries: 1024 * (2 * FETCH_SIZE + WRITE_SIZE) over the tile-pass kernel + bal_reduce_partials_kernel
(+ bal_camera_chunk_kernel where the cameras do not fit in LDS); FETCH_SIZE doubled per the guide's gfx950 correction.

bench.py calls measure() for roofline.traffic (N = 1, default on); on any failure it falls back to the committed constant
(profiles/pmc_traffic.json) and says so.  The passes never combine --pmc with a trace domain other than --kernel-trace.

usage: tools/pmc_live.py [workload]      prints the JSON measure() returns
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPERATORS = {"sx": 0, "jtjx": 1}


def under_a_profiler():
    """True when this process already runs under rocprofv3 / rocprof (a nested counter pass would fight over the tool library)."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in pre:
        return True
    return any(k.startswith(("ROCPROF_", "ROCPROFILER_", "ROCP_TOOL")) for k in os.environ)


def _counter_means(csv_path, counter):
    agg = collections.defaultdict(list)
    with open(csv_path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def _one_pass(counter, workload, timeout, env_extra):
    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        raise RuntimeError("rocprofv3 not on PATH")
    d = tempfile.mkdtemp(prefix=f"pmc_live_{counter}_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    env.update(env_extra or {})
    cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "tools", "kernel_times.py"), workload, "--operators-only"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise RuntimeError(f"no counter_collection.csv (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}")
        return _counter_means(files[0], counter)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure(workload="venice1778", timeout=300, env_extra=None):
    """{"sx": bytes, "jtjx": bytes, "breakdown_KiB": {...}, "dispatches": {...}, "seconds": s}; raises on failure."""
    t0 = time.time()
    fetch, nf = _one_pass("FETCH_SIZE", workload, timeout, env_extra)
    write, _ = _one_pass("WRITE_SIZE", workload, timeout, env_extra)
    out = {"breakdown_KiB": {}, "dispatches": {}}
    for op, mode in OPERATORS.items():
        tile = None
        for pat in (rf"bal_stream_kernel<{mode}, true(, (true|false)){{0,2}}>", rf"bal_stream_kernel<{mode}, false(, (true|false)){{0,2}}>",
                    rf"bal_fused_kernel<{mode}, (true|false), \d+(, (true|false))?>"):
            if any(re.search(pat, k) for k in fetch):
                tile = pat
                break
        if tile is None:
            continue
        kernels = [tile, r"bal_reduce_partials_kernel"]
        if re.search(rf"<{mode}, false", tile):
            kernels.append(r"bal_camera_chunk_kernel")
        total, brk = 0.0, {}
        for kp in kernels:
            f = sum(v for k, v in fetch.items() if re.search(kp, k))
            w = sum(v for k, v in write.items() if re.search(kp, k))
            if f == 0 and w == 0:
                continue
            brk[kp.replace("\\", "")] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1)}
            total += 1024.0 * (2.0 * f + w)
            out["dispatches"][kp.replace("\\", "")] = sum(n for k, n in nf.items() if re.search(kp, k))
        out[op] = int(round(total))
        out["breakdown_KiB"][op] = brk
    if "sx" not in out and "jtjx" not in out:
        raise RuntimeError("the counter passes saw none of the operators' kernels: " + ", ".join(sorted(fetch)[:6]))
    out["seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    print(json.dumps(measure(sys.argv[1] if len(sys.argv) > 1 else "venice1778"), indent=1))
