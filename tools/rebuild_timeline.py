"""Timeline of ONE list rebuild out of a rocprofv3 --kernel-trace (+ --memory-copy-trace) capture: the kernels and copies
between the last k_nonbond before a rebuild and the first one after it, with the idle gaps between them.
   python3 tools/rebuild_timeline.py <dir with *_kernel_trace.csv> [which rebuild, default: the last]
A steady step as well: the launches between the two k_nonbond launches five steps before that rebuild."""
import csv, glob, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:24] ))
ev.sort()
idx = [i for i, e in enumerate(ev) if "k_tile_build" in e[2]]
if not idx: sys.exit("no k_tile_build in the trace")
i = idx[which]
lo = i
while lo > 0 and "k_nonbond" not in ev[lo][2]: lo -= 1
hi = i
while hi < len(ev) - 1 and "k_nonbond" not in ev[hi][2]: hi += 1
t0 = ev[lo][1]
print("rebuild window: %.1f us from the end of the previous k_nonbond to the start of the next" % ((ev[hi][0] - t0) / 1e3))
busy = 0; last = t0; gaps = 0
for s, e, n in ev[lo + 1:hi + 1]:
    gap = (s - last) / 1e3
    if n != ev[hi][2] or s != ev[hi][0]: busy += (e - s) / 1e3
    if gap > 0: gaps += gap
    print("  +%8.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
    last = max(last, e)
print("busy %.1f us, idle %.1f us" % (busy, gaps))

# one steady-state step: from the start of the fifth pair kernel before the rebuild to the start of the next one
nb = [k for k in range(lo + 1) if "k_nonbond" in ev[k][2]]
if len(nb) > 6:
    a, b = nb[-6], nb[-5]
    print("steady step: %.1f us from the start of one pair kernel to the start of the next" % ((ev[b][0] - ev[a][0]) / 1e3))
    last = ev[a][0]
    for s_, e_, n in ev[a:b + 1]:
        print("  +%8.1f us  gap %7.1f  dur %8.1f  %s" % ((s_ - ev[a][0]) / 1e3, (s_ - last) / 1e3, (e_ - s_) / 1e3, n))
        last = max(last, e_)
