"""Developer timing of the Hamming kernels (not the judged bench): python tools/dev_time_match.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nclt_slam_project_amd.engine import Engine

e = Engine(0, 1280, 720, 8192)
rng = np.random.default_rng(0)
for L, n, Q in [(10000, 64, 500), (10000, 45, 500), (10000, 100, 500), (1000, 64, 500), (100000, 64, 500)]:
    T = L * n
    db = rng.integers(0, 256, (T, 32), dtype=np.uint8)
    off = np.arange(L + 1, dtype=np.int64) * n
    cur = rng.integers(0, 256, (Q, 32), dtype=np.uint8)
    e.db_upload(db, np.zeros((T, 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
    dcur = e.to_device(cur); dcnt = e.dev_alloc(L * 4)
    for _ in range(3): e.db_match_counts_dev(dcur, Q, dcnt)
    e.sync()
    K = 20
    e.timer_begin()
    for _ in range(K): e.db_match_counts_dev(dcur, Q, dcnt)
    ms = e.timer_end() / K
    pairs = T * Q
    print(f"db_scan L={L} n={n} Q={Q}: {ms*1e3:.1f} us  pairs/s={pairs/ms/1e9*1e3:.1f} G  algGB/s={(32*T+32*Q+4*L)/ms/1e6:.1f}", flush=True)
    e.dev_free(dcur); e.dev_free(dcnt)
for F, K_ in [(4096, 4096), (20000, 20000)]:
    a = rng.integers(0, 256, (F, 32), dtype=np.uint8); b = rng.integers(0, 256, (K_, 32), dtype=np.uint8)
    da, db_ = e.to_device(a), e.to_device(b); dout = e.dev_alloc(F * K_ * 2)
    for _ in range(2): e.hamming_matrix_dev(da, F, db_, K_, dout)
    e.sync()
    R = 10
    e.timer_begin()
    for _ in range(R): e.hamming_matrix_dev(da, F, db_, K_, dout)
    ms = e.timer_end() / R
    byts = 32 * (F + K_) + 2 * F * K_
    print(f"matrix {F}x{K_}: {ms*1e3:.1f} us  {byts/ms/1e6:.1f} GB/s ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)  pairs/s={F*K_/ms/1e6:.1f} G", flush=True)
