"""Overlap evidence for the multi-stream bench run (VERDICT r1 "missing" item 3).

Reads a rocprofv3 --kernel-trace CSV of `python bench.py` (default: 16 streams, hipGraph replay) and prints, for the
steady-state window (the middle 60 % of the traced time):
  * per kernel: calls, average duration, share of the summed kernel time;
  * the concurrency histogram: fraction of wall time with k kernels in flight, average kernels in flight;
  * for every kernel family, the average number of FPS kernels (serial chains, one CU per frame) in flight while it
    runs -- the overlap that turns the 5 ms single-stream latency into ~1 ms per step;
  * wall time per step = window / steps inside it, against the sum of kernel durations per step.
usage: python tools/concurrency.py <kernel_trace.csv> [frames_per_step]
"""
import bisect
import collections
import csv
import sys


def short(n):
    for p in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        n = n.replace(p, "")
    return n.split("(")[0][:64]


def main(path, frames_per_step=8):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = hi - lo
    print("trace: %d kernels over %.2f ms; steady-state window %.2f ms with %d kernels" % (len(rows), (t1 - t0) / 1e6, span / 1e6, len(win)))
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, k in win:
        per[k][0] += 1
        per[k][1] += e - s
    tot = sum(v[1] for v in per.values())
    is_fps = lambda k: k.startswith("fps")
    # layer-1 D-FPS launches = steps in the window
    l1 = [k for k in per if "wave_bucket" in k or "fps3_reg_kernel<16>" in k]
    steps = sum(per[k][0] for k in l1) or 1
    print("steps in window (layer-1 FPS launches): %d -> %.4f ms wall per step (%.0f frames/s at %d frames/step); "
          "sum of kernel durations per step %.3f ms" % (steps, span / steps / 1e6, frames_per_step * steps / (span / 1e9), frames_per_step, tot / steps / 1e6))
    # events for concurrency
    ev = []
    for s, e, k in win:
        ev.append((s, 1, is_fps(k)))
        ev.append((e, -1, is_fps(k)))
    ev.sort()
    hist = collections.Counter()
    cur = curf = 0
    last = ev[0][0]
    times = [ev[0][0]]
    fps_level = [0]
    for t, d, f in ev:
        hist[cur] += t - last
        last = t
        cur += d
        if f:
            curf += d
        times.append(t)
        fps_level.append(curf)
    busy = sum(hist.values())
    print("kernels in flight (fraction of window): " + "  ".join("%d:%.3f" % (k, v / busy) for k, v in sorted(hist.items()) if v / busy >= 0.005))
    print("average kernels in flight: %.2f" % (sum(k * v for k, v in hist.items()) / busy))

    def fps_during(s, e):   # time-average of the FPS level over [s, e]
        i = bisect.bisect_right(times, s) - 1
        acc = 0.0
        t = s
        while t < e and i < len(times):
            nt = min(e, times[i + 1]) if i + 1 < len(times) else e
            acc += fps_level[i] * (nt - t)
            t = nt
            i += 1
        return acc / max(e - s, 1)
    print("%-64s %6s %10s %7s %s" % ("kernel", "calls", "avg us", "share", "FPS kernels in flight while it runs"))
    for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        ov = [fps_during(s, e) for s, e, kk in win if kk == k][:400]
        print("%-64s %6d %10.1f %6.1f%% %6.1f" % (k, n, d / n / 1e3, 100.0 * d / tot, sum(ov) / len(ov) - (1 if is_fps(k) else 0)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
