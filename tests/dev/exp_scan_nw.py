"""Developer experiment (round 3): waves per record of the whole-database scan (RELOC_SCAN_NW = 4 / 2 / 1) on fixed and ragged
databases, each checked against the CPU oracle first; one subprocess per setting.
    python tests/dev/exp_scan_nw.py            # single-stream kernel times, default grid and one-generation (exclusive) grid
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = (("fixed64_10k", "fixed64", 10000), ("ragged_10k", "ragged", 10000), ("45_10k", 45, 10000), ("100_10k", 100, 10000),
          ("fixed64_100k", "fixed64", 100000), ("ragged_100k", "ragged", 100000), ("45_100k", 45, 100000))


def one():
    import numpy as np
    from nclt_slam_project_amd.engine import Engine
    from nclt_slam_project_amd import synth
    from oracle import oracle as O
    O.build()
    e = Engine(0, 640, 480, 4096)
    if os.environ.get("EXP_EXCLUSIVE") == "1":
        e.set_exclusive(True)
    rng = np.random.default_rng(3)
    ok = True
    for rows, L, Q in (("ragged", 600, 500), (64, 300, 500), (7, 50, 37), (100, 320, 300), ("ragged", 300, 1000), (45, 500, 500),
                       (17, 400, 480), (1, 300, 500), (33, 1100, 512), (130, 200, 500)):
        base = synth.random_descriptors(rng, Q)
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows, base, planted_records=(3, L // 2))
        desc[off[1]:off[2], 4:] = 0                       # low-entropy rows force distance ties
        e.db_upload(desc, pts, off, poses)
        got = e.db_match_counts(base)
        exp = O.db_match_counts(desc, off, base)
        if not (got == exp).all():
            ok = False
            print(json.dumps(dict(MISMATCH=[str(rows), L, Q], n_bad=int((got != exp).sum()))), flush=True)
    res = dict(nw=os.environ.get("RELOC_SCAN_NW", "default"), exclusive=os.environ.get("EXP_EXCLUSIVE", "0"), parity=ok)
    Q = 500
    for name, rows, L in SHAPES:
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows)
        e.db_upload(desc, pts, off, poses)
        T = int(off[-1])
        cnt = e.dev_alloc(L * 4)
        cur = e.to_device(synth.random_descriptors(rng, Q))
        for _ in range(30):
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
        res[name] = [round(best, 1), round(T * Q / (best * 1e-6) / 1e12, 3)]
        e.dev_free(cur); e.dev_free(cnt)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--one":
        one()
    else:
        for rnd in range(int(os.environ.get("EXP_ROUNDS", "1"))):
            for excl in ("0", "1"):
                for nw in os.environ.get("EXP_NWS", "4,2,1").split(","):
                    env = dict(os.environ, RELOC_DEV="1", RELOC_SCAN_NW=nw, EXP_EXCLUSIVE=excl)
                    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, timeout=900)
