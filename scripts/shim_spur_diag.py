import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from refcases import spur_case, harness_args
d, sp, iq, lim = spur_case("spur_n10_fft1")
td = tempfile.mkdtemp()
iq.tofile(td + "/in.bin"); lim.tofile(td + "/lim.bin")
cmd = [os.path.join(ROOT, "oracle/_ref/shim_harness_hip")] + harness_args(d, td + "/in.bin", td + "/lim.bin", td + "/out.bin") + ["spur=1"] + [f"{k}={v}" for k, v in sp.items()]
r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, LRH_VERBOSE="1"))
print(r.returncode, r.stdout[-500:], r.stderr[-1500:])
