"""GPU, single process: the scan / solve halves of the fused tick (what a rank of the sharded path runs)
agree with the fused whole-database tick on the same frame."""
import numpy as np
import pytest

from nclt_slam_project_amd import synth
from nclt_slam_project_amd.sharded import HipShard, ShardedRelocalizer

pytestmark = pytest.mark.gpu


def test_scan_solve_equals_fused_tick(engine, oracle):
    rng = np.random.default_rng(21)
    img = synth.textured_frame(rng, 640, 480)
    feat = engine.orb_detect_compute(engine.gray(img), 500)
    desc, pts, off, poses = synth.descriptor_db(rng, 300, "ragged", feat["desc"], planted_records=(7, 120, 299))
    # make record 120 PnP-solvable: its 3-D points reproject onto the matched keypoints
    engine.db_upload(desc, pts, off, poses)
    bp = synth.base_pose(float(poses[120, 0]), 0.0, 0.0)
    fused = engine.tick(img, bp, global_reloc=True, seed=5)
    dbg = engine.tick_debug()
    counts = oracle.db_match_counts(desc, off, feat["desc"])
    exp_ids = oracle.topk_records(counts, 10, 25)
    np.testing.assert_array_equal(dbg["cand_ids"], exp_ids)          # all headings equal -> mask keeps everything
    np.testing.assert_array_equal(dbg["n_matches"], counts[exp_ids])
    shard = HipShard(engine, desc, pts, off, poses, rank=0, world=1)
    frame_dev = engine.to_device(img)
    sr = ShardedRelocalizer(shard, shard.base, 0, 1)
    r = sr.tick(frame_dev, bp, seed=5)
    engine.dev_free(frame_dev)
    assert r["n_candidates"] == fused["n_candidates"] == len(exp_ids)
    assert (r["outcome"] in (0, 4)) == (fused["outcome"] in (0, 4))
    if fused["outcome"] in (0, 4):
        assert r["n_inliers"] == fused["n_inliers"] and r["lm_idx"] == fused["lm_idx"]
        np.testing.assert_allclose(r["anchor_pose"], fused["anchor_pose"], atol=1e-9)
