#!/usr/bin/env python3
"""One steady-state round of the default bench run as a timeline, from a rocprofv3 --kernel-trace CSV:
   scripts/timeline.py <dir with *_kernel_trace.csv> [name of the kernel that starts a round] [which launch]
Columns: start / end / duration in us relative to the round's first kernel, stream id, kernel.  The gaps between the main stream's
kernels are the event packets the queue works off."""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_fft1v<14, false, false"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 26
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:72], r.get("Stream_Id")) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
if len(idx) <= nth + 1:
    sys.exit("only %d launches of %s" % (len(idx), anchor))
i0, i1 = idx[nth], idx[nth + 1]
t0 = ks[i0][0]
print("# %s: launch %d of %s to the next one" % (os.path.basename(f), nth, anchor))
print("#   start      end      dur  stream  kernel")
for k in ks[i0:i1 + 1]:
    print("%9.1f %8.1f %8.1f  s=%s  %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
main = [k for k in ks[i0:i1 + 1] if k[3] == ks[i0][3]]
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(main, main[1:])]
print("# main stream: busy %.1f us, gaps %.1f us in %d places (largest %.1f)" % (sum((k[1] - k[0]) / 1e3 for k in main[:-1]), sum(gaps), len(gaps), max(gaps) if gaps else 0))
per = np.diff([ks[i][0] for i in idx[max(nth - 10, 0):nth + 10]]) / 1e3
print("# round period over the neighbouring rounds: median %.1f us (min %.1f, max %.1f)" % (np.median(per), per.min(), per.max()))
