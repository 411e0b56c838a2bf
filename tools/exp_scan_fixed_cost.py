import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from nclt_slam_project_amd.engine import Engine
from nclt_slam_project_amd import synth
e = Engine(0, 640, 480, 2048)
rng = np.random.default_rng(3)
for Q in (500, 128, 65):
    cur = e.to_device(synth.random_descriptors(rng, Q))
    for L, rows in ((1024, 1), (1024, 16), (1024, 64), (2048, 1), (10000, 1), (256, 64), (64, 64)):
        desc, p3, off, poses = synth.descriptor_db(rng, L, rows)
        e.db_upload(desc, p3, off, poses)
        cnt = e.dev_alloc(L * 4)
        for _ in range(40): e.db_match_counts_dev(cur, Q, cnt)
        e.sync()
        best = 1e9
        for rep in range(3):
            e.profile_enable(True)
            for _ in range(40): e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            ms, k = e.profile_get(0); e.profile_enable(False)
            best = min(best, ms / k * 1e3)
        e.dev_free(cnt)
        print(json.dumps(dict(Q=Q, records=L, rows=rows, us=round(best, 1))), flush=True)
    e.dev_free(cur)
