"""Developer experiment: time build variants of libreloc_hip.so (nclt-slam-project_amd/build.py build_variant) on the
whole-database scan shapes, after checking each against the CPU oracle.  One subprocess per library (RELOC_LIB).
    python tests/dev/exp_scan_variants.py [lib.so ...]        # default: csrc/libreloc_hip.so + build_variants/*.so
"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(lib):
    import numpy as np
    from nclt_slam_project_amd.engine import Engine
    from nclt_slam_project_amd import synth
    from oracle import oracle as O
    O.build()
    e = Engine(0, 640, 480, 4096)
    rng = np.random.default_rng(3)
    ok = True
    for rows, L, Q in (("ragged", 300, 500), (64, 200, 500), (7, 50, 37), (100, 120, 300), ("ragged", 200, 1000)):
        base = synth.random_descriptors(rng, Q)
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows, base, planted_records=(3, L // 2))
        # low-entropy rows force distance ties
        desc[off[1]:off[2], 4:] = 0
        e.db_upload(desc, pts, off, poses)
        got = e.db_match_counts(base)
        exp = O.db_match_counts(desc, off, base)
        if not (got == exp).all():
            ok = False
            print(json.dumps(dict(lib=os.path.basename(lib), MISMATCH=[str(rows), L, Q], n_bad=int((got != exp).sum()))), flush=True)
    res = dict(lib=os.path.basename(lib), parity=ok)
    for name, rows, L, Q in (("fixed64_10k_Q500", "fixed64", 10000, 500), ("ragged_10k_Q500", "ragged", 10000, 500),
                             ("fixed64_100k_Q500", "fixed64", 100000, 500), ("fixed64_10k_Q32", "fixed64", 10000, 32),
                             ("fixed64_10k_Q1", "fixed64", 10000, 1)):
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows)
        e.db_upload(desc, pts, off, poses)
        T = int(off[-1])
        cnt = e.dev_alloc(L * 4)
        cur = e.to_device(synth.random_descriptors(rng, Q))
        for _ in range(5):
            e.db_match_counts_dev(cur, Q, cnt)
        e.sync()
        best = 1e9
        for rep in range(3):
            e.profile_enable(True)
            for _ in range(20):
                e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            ms, k = e.profile_get(0)
            e.profile_enable(False)
            best = min(best, ms / k * 1e3)
        res[name + "_us"] = round(best, 1)
        res[name + "_Tpairs"] = round(T * Q / (best * 1e-6) / 1e12, 3)
        e.dev_free(cur); e.dev_free(cnt)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "nclt-slam-project_amd", "csrc", "libreloc_hip.so")] + \
            sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so")))
        for lib in libs:
            for grid in os.environ.get("EXP_GRIDS", "0").split(","):        # RELOC_SCAN_GRID values: 0 ticket, -1 static default
                env = dict(os.environ, RELOC_DEV="1", RELOC_LIB=os.path.abspath(lib), RELOC_SCAN_GRID=grid)
                print(json.dumps(dict(lib=os.path.basename(lib), RELOC_SCAN_GRID=grid)), flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib], env=env, timeout=600)
