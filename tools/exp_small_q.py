"""Developer experiment (rounds 3-4): the few-query scan (k_db_scan_rows) of several library builds in ONE run, rounds interleaved
(boxes of the pool differ by up to 10 %, so only same-run comparisons count).  Round 4 also switched kernel forms inside one build
(EXP_FORMS -> RELOC_SQ_FORM; the switch was removed with the round-3 kernel, profiles/r4_small_q_forms.log): the first build / form
of a round saves its counts and the others are compared with them on every database and query count before they are timed.
    python tools/exp_small_q.py [lib.so ...]      # default: csrc/libreloc_hip.so + build_variants/*.so
"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    from nclt_slam_project_amd.engine import Engine
    from nclt_slam_project_amd import synth
    e = Engine(0, 640, 480, 4096)
    rng = np.random.default_rng(11)
    res = dict(lib=os.path.basename(os.environ.get("RELOC_LIB", "product")), form=os.environ.get("RELOC_SQ_FORM", "0"))
    ref = None
    if res["form"] != "0":                       # a second context of this process cannot differ in form (read at creation): the
        ref = {}                                 # reference counts come from a child that runs form 0 and saves them
    qs = [int(x) for x in os.environ.get("EXP_QS", "1,4,8,16,32").split(",")]
    dbs = (("fixed64_100k", "fixed64", 100000), ("ragged_100k", "ragged", 100000), ("fixed64_10k", "fixed64", 10000))
    for name, rows, L in dbs:
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows)
        e.db_upload(desc, pts, off, poses)
        T = int(off[-1])
        cnt = e.dev_alloc(L * 4)
        for Q in qs:
            cur = e.to_device(synth.random_descriptors(rng, Q))
            e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            got = np.empty(L, np.int32); e.d2h(got, cnt)
            path = os.path.join(os.environ.get("EXP_REF_DIR", "/tmp"), f"sq_ref_{name}_Q{Q}.npy")
            if res["form"] == "0":
                np.save(path, got)
            elif os.path.exists(path):
                res.setdefault("parity", True)
                res["parity"] = bool(res["parity"] and np.array_equal(got, np.load(path)))
            for _ in range(60):
                e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            best = 1e9
            for rep in range(3):
                e.profile_enable(True)
                for _ in range(40):
                    e.db_match_counts_dev(cur, Q, cnt)
                e.sync()
                ms, k = e.profile_get(0)
                e.profile_enable(False)
                best = min(best, ms / k * 1e3)
            res[f"{name}_Q{Q}"] = [round(best, 1), round((32 * T + 32 * Q + 4 * L) / (best * 1e-6) / 8e12, 3)]
            e.dev_free(cur)
        e.dev_free(cnt)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--one":
        one()
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "nclt-slam-project_amd", "csrc", "libreloc_hip.so")] + \
            sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so")))
        forms = os.environ.get("EXP_FORMS", "0").split(",")
        for rnd in range(int(os.environ.get("EXP_ROUNDS", "2"))):
            for lib in libs:
                for form in forms:
                    env = dict(os.environ, RELOC_DEV="1", RELOC_LIB=os.path.abspath(lib), RELOC_SQ_FORM=form)
                    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, timeout=600)
