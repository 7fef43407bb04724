"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel durations, how the streams overlap, and the idle gaps of the GPU.
usage: python scripts/timeline.py <dir with *_kernel_trace.csv> [skip_frames]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].replace("mv::", ""), r.get("Queue_Id", "?")) for r in rows)
# keep the steady state: from the (skip)th marchViewKernel on
mv = [i for i, e in enumerate(ev) if e[2].startswith("marchTiles")]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.34
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.99
ev = ev[mv[int(lo * len(mv))]:mv[min(len(mv) - 2, int(hi * len(mv)))]]
t0, t1 = ev[0][0], max(e[1] for e in ev)
import os
n = sum(1 for e in ev if e[2].startswith("cullKernel")) or int(os.environ.get("TIMELINE_FRAMES_PER_LAUNCH", "4")) * sum(1 for e in ev if e[2].startswith("cullBatchKernel"))  # one cull per frame (the view march may have an overflow launch behind it); several frames per launch: TIMELINE_FRAMES_PER_LAUNCH (4) frames per cull launch
print("frames %d  span %.3f ms  -> %.4f ms/frame" % (n, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / n))
dur = collections.defaultdict(list)
for s, e, k, q in ev: dur[(k, q)].append(e - s)
for (k, q), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s queue %-3s n %5d  avg %8.1f us  per-frame %7.1f us" % (k, q, len(v), sum(v) / len(v) / 1e3, sum(v) / n / 1e3))
# coverage: time with 0, 1, 2+ kernels running
pts = []
for s, e, k, q in ev: pts += [(s, 1), (e, -1)]
pts.sort()
cov = collections.Counter(); cur = 0; last = pts[0][0]
for t, dlt in pts:
    cov[min(cur, 3)] += t - last; last = t; cur += dlt
tot = sum(cov.values())
print("  concurrency: " + "  ".join("%d kernels: %.1f%% (%.1f us/frame)" % (k, 100.0 * v / tot, v / n / 1e3) for k, v in sorted(cov.items())))
# a sample of the steady-state schedule
base = ev[len(ev) // 2][0]
for s, e, k, q in ev[len(ev) // 2: len(ev) // 2 + 40]:
    print("   %9.1f .. %9.1f us  q%-3s %s" % ((s - base) / 1e3, (e - base) / 1e3, q, k))
# the largest idle gaps (no kernel running) of the window
gaps = []; cur = 0; last = None
for t, dlt in pts:
    if cur == 0 and last is not None and t - last > 30000: gaps.append((t - last, last))
    cur += dlt
    if cur == 0: last = t
print("  idle gaps > 30 us: %d, total %.1f us/frame" % (len(gaps), sum(g for g, _ in gaps) / n / 1e3))
for g, at in sorted(gaps, reverse=True)[:8]:
    prev = [e for e in ev if e[1] <= at][-3:]; nxt = [e for e in ev if e[0] >= at + g][:2]
    print("   gap %.1f us at %.1f us after %s before %s" % (g / 1e3, (at - t0) / 1e3, [p[2] for p in prev], [q[2] for q in nxt]))
# per queue: how long a queue sits between the end of one kernel and the start of the next (by the kernel that follows), and the queue's cycle
print("  per-queue gaps in front of a kernel (mean us over the window):")
byq = collections.defaultdict(list)
for s, e, k, q in ev: byq[q].append((s, e, k))
for q, lst in sorted(byq.items()):
    lst.sort()
    g = collections.defaultdict(list)
    for (s0, e0, k0), (s1, e1, k1) in zip(lst, lst[1:]): g[(k0, k1)].append((s1 - e0) / 1e3)
    busy = sum(e - s for s, e, _ in lst) / 1e3
    print("   q%-3s busy %.1f us/frame, span %.1f us/frame:" % (q, busy / n, (lst[-1][1] - lst[0][0]) / 1e3 / n) + "".join("  %s->%s %.1f" % (a[:10], b[:10], sum(v) / len(v)) for (a, b), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:8]))
# the steady-state schedule as lanes: every kernel of 2.5 ms from the middle of the window, one column per queue
if len(sys.argv) > 4 or True:
    base = ev[len(ev) // 2][0]
    qs = sorted(byq)
    print("  lanes (start us, duration us; one column per queue):")
    for s, e, k, q in ev[len(ev) // 2:]:
        if s - base > 2500e3: break
        col = qs.index(q)
        print("   %8.1f  " % ((s - base) / 1e3) + " " * (34 * col) + "%-22s %6.1f" % (k[:22], (e - s) / 1e3))
