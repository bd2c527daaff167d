"""diagnostic: per-block weak / strong differences between the full-ring and the sparse run of tests/test_gpu_fused.py's gap case"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_fused as t

for table in (0, 1):
    a, la = t._run_gap(None, 0, table)
    sp, ls = t._run_gap(None, 1, table)
    n1 = 1 << 14
    blk = 4 * (n1 // 2)
    print("table", table, la, ls)
    for b in (0, 1, 30, 31, 32, 33, 34, 63, 64, 65, 95):
        x = a["timf2"][b * blk:(b + 1) * blk].reshape(-1, 4).astype(np.float64)
        y = sp["timf2"][b * blk:(b + 1) * blk].reshape(-1, 4).astype(np.float64)
        w = np.linalg.norm(x[:, :2] - y[:, :2]) / max(np.linalg.norm(x[:, :2]), 1e-30)
        s = np.linalg.norm(x[:, 2:] - y[:, 2:]) / max(np.linalg.norm(x[:, 2:]), 1e-30)
        print("  block %3d weak %.3e (|a| %.3e |sp| %.3e)  strong %.3e (|a| %.3e)" % (b, w, np.linalg.norm(x[:, :2]), np.linalg.norm(y[:, :2]), s, np.linalg.norm(x[:, 2:])))
