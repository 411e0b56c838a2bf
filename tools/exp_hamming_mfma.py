"""OFF-CONTRACT micro-benchmark (tools/ only, never part of the product path): what the north star's
"no MFMA for the Hamming stage" clause costs.  Same work as the product's whole-database scan -- per-record mutual-match counts of
10 000 x 64 teach rows against 512 current descriptors -- with the distances from v_mfma_i32_32x32x32_i8 (tools/exp_hamming_mfma.hip),
checked record by record against reloc_db_match_counts (k_db_scan, XOR / popcount on the VALU).
    python tools/exp_hamming_mfma.py build     # in the build container: hipcc -> tools/libexp_hamming_mfma.so (git-ignored, travels)
    python tools/exp_hamming_mfma.py           # on the GPU box
"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libexp_hamming_mfma.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO,
                    os.path.join(ROOT, "tools", "exp_hamming_mfma.hip")], check=True)
    print("built", SO)
    sys.exit(0)

import numpy as np
from nclt_slam_project_amd.engine import Engine
from nclt_slam_project_amd import synth

e = Engine(0, 640, 480, 2048)                 # also puts the process on ONE HIP runtime before the experiment library loads
lib = C.CDLL(SO)
lib.mfma_scan.restype = C.c_float
lib.mfma_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
lib.mfma_expand_b.argtypes = [C.c_void_p, C.c_void_p]
rng = np.random.default_rng(8)
out = []
for L in (10000, 100000):
    Q = 512
    cur = synth.random_descriptors(rng, Q)
    desc, pts, off, poses = synth.descriptor_db(rng, L, "fixed64", cur, planted_records=tuple(rng.choice(L, 20, replace=False)))
    # massive ties too: a few records of low-entropy rows
    for r in rng.choice(L, 20, replace=False):
        desc[off[r]:off[r + 1]] &= 0x11
    e.db_upload(desc, pts, off, poses)
    ref = e.db_match_counts(cur)               # the product's scan (k_db_scan<8,false,4>)
    d_db = e.to_device(desc); d_cur = e.to_device(cur)
    d_img = e.dev_alloc(16 * 9 * 64 * 16); d_cnt = e.dev_alloc(L * 4)
    assert lib.mfma_expand_b(d_cur, d_img) == 0
    ms = lib.mfma_scan(d_db, L, d_img, d_cnt, 256, 20)
    assert ms > 0, ms
    got = np.empty(L, np.int32); e.d2h(got, d_cnt)
    bad = int((got != ref).sum())
    # the product's scan alone, same shape
    cnt2 = e.dev_alloc(L * 4)
    e.set_exclusive(True)
    for _ in range(30):
        e.db_match_counts_dev(d_cur, Q, cnt2)
    e.sync()
    e.profile_enable(True)
    for _ in range(20):
        e.db_match_counts_dev(d_cur, Q, cnt2)
    e.sync()
    pms, k = e.profile_get(0)
    e.profile_enable(False)
    e.set_exclusive(None)
    pairs = L * 64 * Q
    out.append(dict(records=L, rows=64, Q=Q, mismatching_records=bad, mfma_us=round(ms * 1e3, 1), mfma_pairs_per_s=round(pairs / (ms * 1e-3) / 1e12, 2),
                    valu_product_us=round(pms / k * 1e3, 1), valu_product_pairs_per_s=round(pairs / (pms / k * 1e-3) / 1e12, 2),
                    ratio=round((pms / k) / ms, 2)))
    print(json.dumps(out[-1]), flush=True)
    for p in (d_db, d_cur, d_img, d_cnt, cnt2):
        e.dev_free(p)
print(json.dumps(dict(note="T pairs/s; MFMA path: i8 32x32x32, bits expanded on the fly, B resident in LDS, same argmin bookkeeping and tie rule; "
                           "off-contract (BASELINE.json north star: no MFMA for the Hamming stage), not wired into reloc_*")))
