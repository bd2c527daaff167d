"""Summary of scripts/glue_trace.sh's traces (rocprofv3's rocpd database): per kernel / copy kind the count, total and mean duration, the device's busy time
(union of the kernel and copy intervals) against the span of the run's last 80 %, and -- when the trace has them (glue_trace_api.sh) -- the HIP runtime
calls per host thread.  usage: python3 scripts/glue_trace_summary.py gpurun_out/glue_prof0 [...]"""
import glob
import sqlite3
import sys
from collections import defaultdict


def busy(iv, cut):
    b, e = 0, cut
    for s, f in sorted(iv):
        if f <= cut:
            continue
        s = max(s, cut)
        if f > e:
            b += f - max(s, e)
            e = f
    return b


for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*results.db", recursive=True)
    if not f:
        print(d, "no rocpd database")
        continue
    db = sqlite3.connect(f[0])
    ks = db.execute("select name,start,end from kernels order by start").fetchall()
    cs = db.execute("select name,start,end,size from memory_copies order by start").fetchall()
    t0, t1 = ks[0][1], max(k[2] for k in ks)
    cut = t0 + (t1 - t0) // 5
    span = t1 - cut
    kb, cb = busy([(k[1], k[2]) for k in ks], cut), busy([(c[1], c[2]) for c in cs], cut)
    ab = busy([(k[1], k[2]) for k in ks] + [(c[1], c[2]) for c in cs], cut)
    print(f"== {d}: span {span / 1e6:.1f} ms (last 80 % of the run); kernels busy {kb / span:.2f}, copies busy {cb / span:.2f}, either {ab / span:.2f}; "
          f"{sum(1 for k in ks if k[1] >= cut)} kernel launches, {sum(1 for c in cs if c[1] >= cut)} copies")
    per = defaultdict(lambda: [0, 0, 0])
    for n, s, e in ks:
        if s >= cut:
            n = n.split("(")[0].replace("void ", "").replace("lrh::", "")[:44]
            per[n][0] += 1
            per[n][1] += e - s
    for n, s, e, sz in cs:
        if s >= cut:
            n = "copy " + n.replace("MEMORY_COPY_", "").replace("_", " ").lower() + (" < 64 kB" if sz < 65536 else "")
            per[n][0] += 1
            per[n][1] += e - s
            per[n][2] += sz
    print(f"   {'kernel / copy':46s} {'count':>6s} {'total ms':>9s} {'mean us':>8s} {'of span':>8s} {'mean bytes':>11s}")
    for n, (c, t, b) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"   {n:46s} {c:6d} {t / 1e6:9.2f} {t / c / 1e3:8.2f} {t / span:8.3f} {b // max(c, 1) if b else '':>11}")
    try:
        rs = db.execute("select name,tid,start,end from regions where start>=? and start<?", (cut, t1)).fetchall()
    except sqlite3.Error:
        rs = []
    if rs:
        api = defaultdict(lambda: [0, 0])
        for n, tid, s, e in rs:
            api[(tid, n)][0] += 1
            api[(tid, n)][1] += e - s
        for tid in sorted({k[0] for k in api}):
            tot = sum(v[1] for k, v in api.items() if k[0] == tid)
            if tot / span < 0.05:
                continue
            print(f"   host thread {tid}: {tot / span:.2f} of the span inside HIP runtime calls")
            for (t_, n), (c, du) in sorted(((k, v) for k, v in api.items() if k[0] == tid), key=lambda kv: -kv[1][1])[:6]:
                print(f"      {n:30s} {c:6d} calls {du / c / 1e3:8.1f} us each {du / span:7.3f}")
