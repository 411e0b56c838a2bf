"""Developer experiment: the production k_hamming_matrix (through the C-ABI, HIP events on the ctx stream) at 20000 x 20000,
to be run next to tools/exp_matrix2 on the same box.   python tools/exp_matrix_prod.py [lib.so ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    from nclt_slam_project_amd.engine import Engine
    e = Engine(0, 640, 480, 2048)
    F = K = 20000
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, (F, 32), dtype=np.uint8)
    a = e.to_device(d); b = e.to_device(d); out = e.dev_alloc(F * K * 2)
    for _ in range(3):
        e.hamming_matrix_dev(a, F, b, K, out)
    e.sync()
    ts = []
    for r in range(80):
        e.timer_begin()
        for _ in range(4):
            e.hamming_matrix_dev(a, F, b, K, out)
        ts.append(e.timer_end() / 4 * 1e3)
    med = sorted(ts)[len(ts) // 2]
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("RELOC_LIB", "libreloc_hip.so")), median_us=round(med, 1), min_us=round(min(ts), 1),
                          frac_of_8TBps=round(2.0 * F * K / med / 1e6 / 8.0, 4), all_us=[round(t) for t in ts])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--one":
        one()
    else:
        for lib in sys.argv[1:] or [""]:
            env = dict(os.environ)
            if lib:
                env["RELOC_DEV"] = "1"; env["RELOC_LIB"] = os.path.abspath(lib)
            subprocess.run([sys.executable, __file__, "--one"], env=env, check=False)
