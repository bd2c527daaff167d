import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import test_gpu_random_configs as t
from linrad_amd import abi
from paritylib import run_case
from refcases import make_input, make_liminfo
d, batch = t.random_case(9)
g = {"iq": make_input(d), "liminfo": make_liminfo(d)}
a = run_case(t._open_hip, "random", golden=g, batch=batch, params=d)
wa = a["wf_lines"].astype(int)
print("fused env", os.environ.get("LRH_FFT2_FUSED"), "line0[250:262]", wa[0, 250:262].tolist(), "unique tail values", np.unique(wa[:, 256:]).tolist()[:10])
yf = np.zeros(1 << d["n1"], np.float32)
rx = a["api"]
rx._proto("get_table", [abi.C.c_void_p, abi.C.c_char_p, abi.C.POINTER(abi.C.c_float), abi.C.c_int]) if hasattr(rx, "_proto") else None
try:
    rc = rx._f("get_table")(rx.ctx, b"yfac", yf.ctypes.data_as(abi.C.POINTER(abi.C.c_float)), yf.size)
    print("yfac rc", rc, yf[:4], "1000*log10(yfac[0]*x)=-2065 -> x =", 10 ** (-2.065) / yf[0])
except Exception as e:
    print("get_table failed", e)
