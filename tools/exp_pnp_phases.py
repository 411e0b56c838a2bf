"""Developer experiment: where the time of k_pnp_hyp / k_pnp_finish goes.  Builds the library with -DRELOC_PNP_TIMING
(build_variants/, never the product .so), runs reloc_pnp_ransac on synthetic problems and prints the phase boundaries the
first wave stamped with the 100 MHz wall clock.
    python tools/exp_pnp_phases.py            # on the GPU box"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    from nclt_slam_project_amd import _native, synth
    from nclt_slam_project_amd.engine import Engine
    lib = _native.load(strict=False)
    lib.reloc_debug_pnp_phases.argtypes = [C.c_void_p]
    e = Engine(0, 640, 480, 2048)
    rng = np.random.default_rng(5)
    rows = []
    for rep in range(24):
        obj, img, rvec, tvec, inl = synth.pnp_problem(rng, 64, 0.25, 0.5)
        for _ in range(2):
            e.pnp_ransac(obj, img, seed=rep)
        ph = (C.c_ulonglong * 32)()
        assert lib.reloc_debug_pnp_phases(ph) == 0
        t = [int(x) for x in ph]
        us = lambda a, b: (t[b] - t[a]) / 100.0
        rows.append(dict(hyp_sample=us(0, 1), hyp_loads=us(1, 2), hyp_coeff=us(2, 3), hyp_cubic=us(3, 4), hyp_quartic_rest=us(4, 5),
                         hyp_solutions=us(5, 6), hyp_pick_store=us(6, 7), hyp_total=us(0, 7),
                         fin_cap=us(8, 9), fin_inliers=us(9, 10), fin_first_normal=us(10, 11), fin_lm=us(11, 12), fin_tail=us(12, 13),
                         fin_total=us(8, 13), lm_trials=t[16], n_inl=t[14], lm_chol=us(17, 18), lm_exp=us(18, 19), lm_normal=us(19, 20)))
    keys = rows[0].keys()
    med = {k: float(np.median([r[k] for r in rows])) for k in keys}
    print(json.dumps(dict(median_us=med, note="first wave of each kernel, 64 matches, 25 % outliers; hyp_cubic includes the depressed-quartic setup")))
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        import importlib.util
        spec = importlib.util.spec_from_file_location("reloc_build", os.path.join(ROOT, "nclt-slam-project_amd", "build.py"))
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        lib = b.build_variant("pnp_timing", ["RELOC_PNP_TIMING"])
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, RELOC_DEV="1", RELOC_LIB=lib, RELOC_DEV="1"), check=True)
