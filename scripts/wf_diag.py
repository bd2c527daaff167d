import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from linrad_amd import abi
from linrad_amd.lib import open_hip, synth_defaults, synth_iq
from linrad_amd.workload import chain_config, strong_liminfo
N1 = 16384
s = synth_defaults(N1, 0)
def run(batch, sparse, pipeline, persist="1"):
    os.environ["LRH_PERSIST"] = persist
    if pipeline is None: os.environ.pop("LRH_PIPELINE", None)
    else: os.environ["LRH_PIPELINE"] = pipeline
    cfg = chain_config(14, 16, batch=4096, fft3_n=12, mix2_n=8, rounds=2)
    cfg.fft1_float_sparse = cfg.fft2_float_sparse = sparse
    cfg.stupid_bln_mode = 0
    rx = open_hip(cfg)
    rx.timf1_write(synth_iq(s, 0, cfg.timf1_bytes // 4)); rx.set_liminfo(strong_liminfo(s, 14)); rx.set_mix1_selfreq(0.31 * 65536 + 0.3)
    rx.wideband_dsp(2 * 4096, batch)
    wf = rx.export(abi.RING_WG_WATERF); p = rx.p.as_dict(); rx.close()
    return wf, p
cases = {"b4096 lagged sparse": (4096, 1, None), "b4096 lagged full": (4096, 0, None), "b4096 serial full": (4096, 0, "0"), "b4096 lagged sparse nopersist": (4096, 1, None, "0"),
         "b256 serial full": (256, 0, "0"), "b256 lagged full": (256, 0, "2"), "b1024 serial full": (1024, 0, "0")}
res = {k: run(*v) for k, v in cases.items()}
ref = res["b256 serial full"][0]
for k, (wf, p) in res.items():
    d = np.nonzero(wf != ref)[0]
    print(k, "diff vs b256 serial:", d.size, "lines", np.unique(d // 1024)[:12], "wptr", p["wg_waterf_ptr"], "ctr", p["wg_waterf_sum_counter"])
