import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from paritylib import *
from refcases import *
from linrad_amd.lib import open_hip
from oracle_binding import open_oracle
g=load_golden('n13_n15_big2'); d=case_params('n13_n15_big2')
cfg=lrh_config(d,g['iq'])
res=[]
for fn in (open_hip, open_oracle):
    api=fn(cfg); api.timf1_write(g['iq']); api.set_liminfo(g['liminfo'])
    tr=[]
    for b in range(9):
        api.fft1_b(1); api.fft1_c(1); api.make_timf2(1); api.first_noise_blanker()
        bs=api.blanker_state(); tr.append((api.p.timf2p_fit, bs.timf2_despiked_pwrinc[0], bs.last_call_cleared, bs.timf2_noise_floor))
    res.append(tr)
for a,b in zip(*res): print(a,b)
