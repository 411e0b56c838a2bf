"""Profiling target: only the two Hamming kernels, fixed shapes (run under rocprofv3 --pmc ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nclt_slam_project_amd.engine import Engine

e = Engine(0, 640, 480, 2048)
rng = np.random.default_rng(0)
L, n, Q = 10000, 64, 500
T = L * n
db = rng.integers(0, 256, (T, 32), dtype=np.uint8)
off = np.arange(L + 1, dtype=np.int64) * n
cur = rng.integers(0, 256, (Q, 32), dtype=np.uint8)
e.db_upload(db, np.zeros((T, 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
dcur = e.to_device(cur); dcnt = e.dev_alloc(L * 4)
for _ in range(10):
    e.db_match_counts_dev(dcur, Q, dcnt)
e.sync()
# the few-query scan (k_db_scan_rows, lane = teach row) on the 100 000-record database of roofline_small_q, Q = 1, 8 and 32
L2 = 100000
db2 = rng.integers(0, 256, (L2 * n, 32), dtype=np.uint8)
off2 = np.arange(L2 + 1, dtype=np.int64) * n
e2 = Engine(0, 640, 480, 2048)
e2.db_upload(db2, np.zeros((L2 * n, 3), np.float32), off2, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L2, 1)))
dcnt2 = e2.dev_alloc(L2 * 4)
for Q2 in (1, 8, 32):
    dq = e2.to_device(cur[:Q2])
    for _ in range(6):
        e2.db_match_counts_dev(dq, Q2, dcnt2)
    e2.sync()
F = K = 20000
a = e.to_device(rng.integers(0, 256, (F, 32), dtype=np.uint8)); b = e.to_device(rng.integers(0, 256, (K, 32), dtype=np.uint8))
out = e.dev_alloc(F * K * 2)
for _ in range(5):
    e.hamming_matrix_dev(a, F, b, K, out)
e.sync()
print("done")
