"""Record-sharded landmark database: one rank per GPU, `torch.distributed` over RCCL (backend
"nccl" on ROCm) or gloo.

The path shards naturally: a record's descriptors and 3-D points stay on one
rank, so mutual matching and PnP are local.  Per frame there is exactly one real exchange step:
every rank scans ITS shard and produces its local top-k (count, global record id) list (k = 25,
200 bytes); one all-gather makes the lists global, every rank performs the same merge, and each
rank solves PnP only for the winners it owns; a second all-gather of one small result record picks
the anchor.  Both collectives are latency-bound (hundreds of bytes), not xGMI-bandwidth-bound.

A 10k-record database is 20 MB: it fits one MI355X thousands of times over, so sharding buys scan
throughput, not capacity (bench.py shards frames instead; this module is the path for databases
that are scanned faster split, BASELINE.json config 4).
"""
from __future__ import annotations

import numpy as np

from .landmarks import shard_by_rows

K_GLOBAL = 25


def merge_topk(all_ids, all_counts, k=K_GLOBAL):
    """Identical on every rank: the k best (count desc, global id desc) entries of the gathered lists
    (`scored.sort(reverse=True)[:25]` over the whole database, reference G:342-343)."""
    ids = np.asarray(all_ids).reshape(-1)
    cnt = np.asarray(all_counts).reshape(-1)
    keep = ids >= 0
    ids, cnt = ids[keep], cnt[keep]
    order = np.lexsort((-ids, -cnt))
    return ids[order][:k], cnt[order][:k]


class ShardedRelocalizer:
    """backend: an object with
         scan(frame, base_pose, k) -> (local ids (k,), counts (k,)) padded with -1 / 0
         solve(local_ids, base_pose, check_consistency, seed) -> dict(outcome, n_inliers, reproj, anchor_pose, lm_idx)
       (HipShard below wraps an Engine; tests use an oracle-backed double)."""

    def __init__(self, backend, shard_base: int, rank: int = 0, world: int = 1, group=None, device=None):
        self.backend, self.base, self.rank, self.world, self.group, self.device = backend, int(shard_base), rank, world, group, device

    def _all_gather(self, arr: np.ndarray) -> np.ndarray:
        if self.world == 1:
            return arr[None]
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def tick(self, frame, base_pose, seed: int = 0, k: int = K_GLOBAL):
        lids, cnts = self.backend.scan(frame, base_pose, k)
        gids = np.where(lids >= 0, lids + self.base, -1).astype(np.int64)
        packed = np.stack([gids, cnts.astype(np.int64)])                      # (2, k) int64: 400 bytes
        allp = self._all_gather(packed)                                       # (world, 2, k)
        win_ids, win_cnt = merge_topk(allp[:, 0], allp[:, 1], k)
        mine = [(pos, int(g - self.base)) for pos, g in enumerate(win_ids) if self._owns(int(g))]
        res = np.zeros(12, np.float64)                                        # [pos, n_inl, reproj, outcome, gid, pose7]
        res[0] = 1e9; res[3] = 3 if len(win_ids) else 2
        if mine:
            r = self.backend.solve([l for _, l in mine], base_pose, False, seed)
            if r["outcome"] in (0, 4):
                pos = next(p for p, l in mine if l == r["lm_idx"])
                res[:5] = [pos, r["n_inliers"], r["reproj"], r["outcome"], r["lm_idx"] + self.base]
                res[5:] = r["anchor_pose"]
            elif r["outcome"] == 1:
                res[3] = 1
        allr = self._all_gather(res)
        if (allr[:, 3] == 1).any():
            return dict(outcome=1, n_inliers=0, reproj=0.0, anchor_pose=np.zeros(7), lm_idx=-1, n_candidates=len(win_ids))
        ok = allr[:, 1] > 0
        if not ok.any():
            return dict(outcome=int(res[3]), n_inliers=0, reproj=0.0, anchor_pose=np.zeros(7), lm_idx=-1,
                        n_candidates=len(win_ids))
        # most inliers, earliest in the global candidate order on ties (reference M:379)
        best = min(np.nonzero(ok)[0], key=lambda i: (-allr[i, 1], allr[i, 0]))
        b = allr[best]
        return dict(outcome=int(b[3]), n_inliers=int(b[1]), reproj=float(b[2]), anchor_pose=b[5:].copy(), lm_idx=int(b[4]),
                    n_candidates=len(win_ids))

    def _owns(self, gid: int) -> bool:
        return self.base <= gid < self.base + self.backend.n_records


class HipShard:
    """One rank's shard on its GPU.  `frame` is a device pointer to a (H, W, 3) uint8 BGR image."""

    def __init__(self, engine, desc, pts3d, offsets, poses, rank: int, world: int, w=640, h=480):
        bounds = shard_by_rows(offsets, world)
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        off = np.asarray(offsets[a:b + 1], np.int64) - int(offsets[a])
        engine.db_upload(desc[offsets[a]:offsets[b]], pts3d[offsets[a]:offsets[b]], off, poses[a:b])
        self.engine, self.base, self.n_records, self.w, self.h = engine, a, b - a, w, h

    def scan(self, frame_dev, base_pose, k):
        return self.engine.tick_scan(frame_dev, self.w, self.h, base_pose, k)

    def solve(self, local_ids, base_pose, check_consistency, seed):
        return self.engine.tick_solve(local_ids, base_pose, check_consistency, seed)
