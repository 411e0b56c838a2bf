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
MIN_FEATURES = 10      # MIN_MATCHES: fewer current keypoints -> curr_no_features (M:307)


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
         scan(frame, base_pose, k, slot) -> (local ids (k,), counts (k,)) padded with -1 / 0
         solve(local_ids, base_pose, check_consistency, seed, slot)
             -> dict(outcome, n_inliers, reproj, anchor_pose, lm_idx)
       `slot` names which of the batch's frames the call is about (solve works on the features the scan of the same
       slot left behind); optional scan_batch(frames, base_poses, k) / solve_batch(jobs, base_poses, seeds) let a
       backend overlap the frames of a batch (HipShard below does; tests use an oracle-backed double)."""

    def __init__(self, backend, shard_base: int, rank: int = 0, world: int = 1, group=None, device=None):
        self.backend, self.base, self.rank, self.world, self.group, self.device = backend, int(shard_base), rank, world, group, device

    def _all_gather(self, arr: np.ndarray) -> np.ndarray:
        if self.world == 1:
            return arr[None]
        if hasattr(self.group, "all_gather_np"):        # in-process group (tests: several shards on one GPU)
            return self.group.all_gather_np(self.rank, arr)
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def tick(self, frame, base_pose, seed: int = 0, k: int = K_GLOBAL):
        return self.tick_batch([frame], [base_pose], [seed], k)[0]

    def tick_batch(self, frames, base_poses, seeds=None, k: int = K_GLOBAL):
        """B frames, two collectives in total (BASELINE config 4: 8 frames per batch): one all-gather of the B local
        top-k lists (B x 400 bytes per rank), one of the B result records (B x 96 bytes)."""
        B = len(frames)
        seeds = list(seeds) if seeds is not None else [0] * B
        be = self.backend
        if hasattr(be, "scan_batch"):
            scans = be.scan_batch(frames, base_poses, k)
        else:
            scans = [be.scan(frames[i], base_poses[i], k, i) for i in range(B)]
        # per frame: k global ids, k counts, and the frame's feature count (-1 from a rank that did not extract: an empty
        # shard) -- the reference decides `curr_no_features` before any candidate work (M:307-309)
        packed = np.zeros((B, 2 * k + 1), np.int64)
        for i, scan in enumerate(scans):
            lids, cnts = scan[0], scan[1]
            packed[i, :k] = np.where(lids >= 0, lids + self.base, -1)
            packed[i, k:2 * k] = cnts
            packed[i, 2 * k] = scan[2] if len(scan) > 2 else MIN_FEATURES
        allp = self._all_gather(packed)                                       # (world, B, 2k + 1)
        n_feat = allp[:, :, 2 * k].max(axis=0)
        winners, jobs = [], []
        for i in range(B):
            if n_feat[i] < MIN_FEATURES:
                winners.append((np.zeros(0, np.int64), []))
                continue
            win_ids, _ = merge_topk(allp[:, i, :k], allp[:, i, k:2 * k], k)
            mine = [(pos, int(g - self.base)) for pos, g in enumerate(win_ids) if self._owns(int(g))]
            winners.append((win_ids, mine))
            if mine:
                jobs.append((i, [l for _, l in mine]))
        if hasattr(be, "solve_batch"):
            solved = be.solve_batch(jobs, base_poses, seeds)
        else:
            solved = [be.solve(ids, base_poses[i], False, seeds[i], i) for i, ids in jobs]
        res = np.zeros((B, 12), np.float64)                                   # [pos, n_inl, reproj, outcome, gid, pose7]
        for i in range(B):
            res[i, 0] = 1e9
            res[i, 3] = 1 if n_feat[i] < MIN_FEATURES else (3 if len(winners[i][0]) else 2)
        for (i, _), r in zip(jobs, solved):
            mine = winners[i][1]
            if r["outcome"] in (0, 4):
                pos = next(p for p, l in mine if l == r["lm_idx"])
                res[i, :5] = [pos, r["n_inliers"], r["reproj"], r["outcome"], r["lm_idx"] + self.base]
                res[i, 5:] = r["anchor_pose"]
            elif r["outcome"] == 1:
                res[i, 3] = 1
        allr = self._all_gather(res)                                          # (world, B, 12)
        out = []
        for i in range(B):
            ar, n_cand = allr[:, i], len(winners[i][0])
            none = dict(n_inliers=0, reproj=0.0, anchor_pose=np.zeros(7), lm_idx=-1, n_candidates=n_cand)
            if (ar[:, 3] == 1).any():
                out.append(dict(outcome=1, **none))
                continue
            ok = ar[:, 1] > 0
            if not ok.any():
                out.append(dict(outcome=int(res[i, 3]), **none))
                continue
            # most inliers, earliest in the global candidate order on ties (reference M:379)
            best = min(np.nonzero(ok)[0], key=lambda j: (-ar[j, 1], ar[j, 0]))
            b = ar[best]
            out.append(dict(outcome=int(b[3]), n_inliers=int(b[1]), reproj=float(b[2]), anchor_pose=b[5:].copy(),
                            lm_idx=int(b[4]), n_candidates=n_cand))
        return out

    def _owns(self, gid: int) -> bool:
        return self.base <= gid < self.base + self.backend.n_records


class HipShard:
    """One rank's shard on its GPU.  `frame` is a device pointer to a (H, W, 3) uint8 BGR image.  n_slots > 1 keeps
    that many contexts (each with the shard uploaded and its own stream) so the frames of a batch overlap."""

    def __init__(self, engine, desc, pts3d, offsets, poses, rank: int, world: int, w=640, h=480, n_slots: int = 1):
        from .engine import Engine
        bounds = shard_by_rows(offsets, world)
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        off = np.asarray(offsets[a:b + 1], np.int64) - int(offsets[a])
        self.engines = [engine] + [Engine(engine.device, engine.max_w, engine.max_h, engine.max_feat) for _ in range(n_slots - 1)]
        if b > a:
            # one resident copy of the shard; the other slots (streams) scan it through reloc_db_share
            engine.db_upload(desc[offsets[a]:offsets[b]], pts3d[offsets[a]:offsets[b]], off, poses[a:b])
            for e in self.engines[1:]:
                e.db_share(engine)
        self.engine, self.base, self.n_records, self.w, self.h = engine, a, b - a, w, h

    def _empty(self, k):
        return np.full(k, -1, np.int32), np.zeros(k, np.int32), -1

    def scan(self, frame_dev, base_pose, k, slot=0):
        if self.n_records == 0:                      # more ranks than records: this rank only takes part in the exchange
            return self._empty(k)
        return self.engines[slot].tick_scan(frame_dev, self.w, self.h, base_pose, k)

    def solve(self, local_ids, base_pose, check_consistency, seed, slot=0):
        return self.engines[slot].tick_solve(local_ids, base_pose, check_consistency, seed)

    def scan_batch(self, frames_dev, base_poses, k):
        if len(frames_dev) > len(self.engines):
            raise ValueError(f"batch of {len(frames_dev)} frames on a shard with {len(self.engines)} slots")
        if self.n_records == 0:
            return [self._empty(k) for _ in frames_dev]
        for i, f in enumerate(frames_dev):
            self.engines[i].tick_scan_enqueue(f, self.w, self.h, base_poses[i], k)
        return [self.engines[i].tick_scan_fetch(k) for i in range(len(frames_dev))]

    def solve_batch(self, jobs, base_poses, seeds):
        for i, ids in jobs:
            self.engines[i].tick_solve_enqueue(ids, base_poses[i], False, seeds[i])
        return [self.engines[i].tick_result() for i, _ in jobs]

    def close(self):
        for e in self.engines[1:]:
            e.close()
