"""diagnostic: examples/lrh_threads single vs threaded, differences per dumped region (tests/test_gpu_threads.py)"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linrad_amd import abi
exe = os.path.join(ROOT, "examples", "lrh_threads")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "linrad_amd", "csrc"), "example"])
sizes = sys.argv[1:4] or ["12", "13", "6"]
outs = []
for t in (0, 1, 1):
    o = f"/tmp/thr_{t}_{len(outs)}.bin"
    r = subprocess.run([exe, str(t), "700", o, *sizes], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip(), r.stderr.strip())
    outs.append(np.fromfile(o, np.uint8))
n1, n2 = 1 << int(sizes[0]), 1 << int(sizes[1])
psz = ctypes.sizeof(abi.LrhPtrs)
bsz = ctypes.sizeof(abi.LrhBlankerState)
tp = 32 << max(int(sizes[0]), int(sizes[1]))
regions = [("ptrs", psz), ("bs", bsz), ("fft1", 64 * 2 * n1 * 4), ("sumsq", None), ("slowsum", n1 * 4), ("timf2", 4 * tp * 4), ("pwr", tp * 4), ("fft2", 16 * 2 * n2 * 4),
           ("powersum", n2 * 4), ("wf", None), ("timf3", (1 << 16) * 4)]
for k in (1, 2):
    a, b = outs[0], outs[k]
    print("run", k, "size", a.size, b.size, "diff bytes", int((a != b).sum()))
    pa = abi.LrhPtrs.from_buffer_copy(a[:psz].tobytes()); pb = abi.LrhPtrs.from_buffer_copy(b[:psz].tobytes())
    print({f: (getattr(pa, f), getattr(pb, f)) for f, _ in abi.LrhPtrs._fields_ if getattr(pa, f) != getattr(pb, f)})
    bad = np.nonzero(a != b)[0]
    if bad.size:
        h, e = np.histogram(bad, bins=40, range=(0, a.size))
        print("histogram of differing byte offsets (40 bins of %d):" % (a.size // 40), h.tolist())
    off = psz + bsz
    f1 = 64 * 2 * n1 * 4
    x, y = a[off:off + f1].view(np.float32).reshape(64, -1), b[off:off + f1].view(np.float32).reshape(64, -1)
    print("fft1 slots differing:", np.nonzero((x != y).any(axis=1))[0].tolist())
