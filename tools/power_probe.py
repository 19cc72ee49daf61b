#!/usr/bin/env python3
"""Is the engine power-limited?  Two measurements on one MI355X:

 1. socket power and shader clock (hwmon sysfs, polled every ~10 ms; rocm-smi as a fallback at ~2 Hz) while bench.py runs the C1 workload;
 2. the fused up-projection (tld_debug_gemm_bench, 32768 x 3072 x 768, epilogue 6) and a plain 4096^3 GEMM timed as single launches after an idle
    gap, and as 10 / 100 / 1000 back-to-back launches: a kernel that runs faster after idling is limited by an averaged power budget, not by its own
    instruction stream.

    python tools/power_probe.py [--steps 20]
"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "power1_cap", "temp1_input"):
            f = os.path.join(d, name)
            if os.path.exists(f):
                out.setdefault(d, {})[name] = f
    return out


def read(f):
    try:
        with open(f) as h:
            return int(h.read().strip())
    except Exception:
        return None


class Poller(threading.Thread):
    def __init__(self, files, period=0.01):
        super().__init__(daemon=True)
        self.files, self.period, self.rows, self.stop = files, period, [], False

    def run(self):
        t0 = time.time()
        while not self.stop:
            row = [time.time() - t0]
            for d, fs in self.files.items():
                row.append(tuple(read(fs[k]) if k in fs else None for k in ("power1_average", "power1_input", "freq1_input")))
            self.rows.append(row)
            time.sleep(self.period)


def summarize(rows, tag):
    if not rows:
        print(f"{tag}: no samples")
        return
    ncard = len(rows[0]) - 1
    for c in range(ncard):
        pw = [r[1 + c][0] if r[1 + c][0] is not None else r[1 + c][1] for r in rows]
        fq = [r[1 + c][2] for r in rows]
        pw = [p / 1e6 for p in pw if p is not None]
        fq = [f / 1e6 for f in fq if f is not None]
        if pw:
            s = sorted(pw)
            print(f"{tag} card{c}: power W  n={len(pw)} min {s[0]:.0f} median {s[len(s) // 2]:.0f} p90 {s[int(len(s) * .9)]:.0f} max {s[-1]:.0f}")
        if fq:
            s = sorted(fq)
            print(f"{tag} card{c}: sclk MHz n={len(fq)} min {s[0]:.0f} median {s[len(s) // 2]:.0f} p90 {s[int(len(s) * .9)]:.0f} max {s[-1]:.0f}")


ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()

files = hwmon_files()
print("hwmon:", {d: sorted(v) for d, v in files.items()})
for d, fs in files.items():
    if "power1_cap" in fs:
        print("power cap W:", read(fs["power1_cap"]) / 1e6)

# ---- 1. bench under the poller
if files:
    p = Poller(files)
    p.start()
    time.sleep(1.0)
    idle = list(p.rows)
    summarize(idle, "idle")
    n0 = len(p.rows)
    r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", str(a.steps), "--warmup", "2", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    rows = p.rows[n0:]
    print(r.stdout.strip()[:200])
    # the timed region is the tail of the process: keep the last steps * 0.21 s
    tl = rows[-1][0]
    run = [x for x in rows if x[0] > tl - a.steps * 0.21 - 0.3 and x[0] < tl - 0.3]
    summarize(run, "C1 generate")
    # time series, 50 ms bins, for the record
    b = {}
    for x in run:
        pw = x[1][0] if x[1][0] is not None else x[1][1]
        if pw is not None:
            b.setdefault(int(x[0] / 0.05), []).append((pw / 1e6, (x[1][2] or 0) / 1e6))
    print("t(s) power(W) sclk(MHz):", " ".join(f"{k * 0.05:.2f}:{sum(v[0] for v in vs) / len(vs):.0f}/{sum(v[1] for v in vs) / len(vs):.0f}" for k, vs in sorted(b.items())[:60]))
    p.stop = True
else:
    for _ in range(3):
        print(subprocess.run(["rocm-smi", "--showpower", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout[-1500:])

# ---- 2. single launches after idling vs back-to-back runs
from transformer_latent_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
ms = C.c_double()
for name, (M, N, K, epi, ntok) in {"up+dwconv 32768x3072x768": (32768, 3072, 768, 6, 256), "plain 4096^3": (4096, 4096, 4096, 0, 256), "down 32768x768x3072": (32768, 768, 3072, 3, 256)}.items():
    res = []
    for iters in (1, 1, 1, 1, 10, 100, 1000, 1, 1):
        time.sleep(0.5)
        _lib.check(L.tld_debug_gemm_bench(M, N, K, epi, ntok, iters, C.byref(ms)), "gemm_bench")
        res.append(f"{iters}x: {ms.value * 1e3:.1f}")
    print(f"{name}: us per launch  " + " | ".join(res), flush=True)
