#!/usr/bin/env python3
"""Poll the GPU's hwmon power / clock sysfs files while a command runs; print min / median / max.   python tools/power_poll.py -- <command...>"""
import glob
import statistics
import subprocess
import sys
import threading
import time

cmd = sys.argv[sys.argv.index("--") + 1:]
pw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
fq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
cap = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap"))
print("files:", pw[:2], fq[:2], cap[:2], flush=True)
samples, stop = [], False


def rd(p):
    try:
        return float(open(p).read())
    except Exception:
        return float("nan")


allp = []


def poll():
    while not stop:
        allp.append([rd(q) / 1e6 for q in pw])
        samples.append((time.perf_counter(), 0.0, 0.0))
        time.sleep(0.02)


t = threading.Thread(target=poll, daemon=True)
t.start()
t0 = time.perf_counter()
p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
stop = True
t.join()
print("rc", p.returncode, "seconds", round(time.perf_counter() - t0, 1), "cap W", rd(cap[0]) / 1e6 if cap else None)
# the card that is ours = the one whose power moved most
if allp:
    spans = [max(r[i] for r in allp) - min(r[i] for r in allp) for i in range(len(pw))]
    k = spans.index(max(spans))
    print("cards:", len(pw), "power span per card W:", [round(x) for x in spans], "-> card index", k, pw[k])
    fk = fq[k] if k < len(fq) else None
    samples = [(samples[i][0], allp[i][k], 0.0) for i in range(len(allp))]
w = [s[1] for s in samples if s[1] == s[1]]
f = [s[2] for s in samples if s[2] == s[2]]
if w:
    print(f"power W: n={len(w)} min {min(w):.0f} median {statistics.median(w):.0f} p90 {sorted(w)[int(.9 * len(w))]:.0f} max {max(w):.0f}")
    top = sorted(samples, key=lambda s: -s[1])[:5]
    print("top samples (t, W, MHz):", [(round(a - t0, 2), round(b), round(c)) for a, b, c in top])
if f:
    print(f"sclk MHz: min {min(f):.0f} median {statistics.median(f):.0f} max {max(f):.0f}")
# the busy window: samples above 60 % of the maximum
if w:
    busy = [s for s in samples if s[1] > 0.6 * max(w)]
    if busy:
        print(f"busy window: {len(busy)} samples, power median {statistics.median([s[1] for s in busy]):.0f} W, sclk median {statistics.median([s[2] for s in busy]):.0f} MHz")
print(p.stdout.strip().splitlines()[-1][:300] if p.stdout.strip() else "")
