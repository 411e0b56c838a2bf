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

Load order: PyTorch-ROCm ships its own HIP runtime.  `import torch` must happen before the first Engine is created
(before libreloc_hip.so loads /opt/rocm's runtime), otherwise torch reports "No HIP GPUs are available" when it
initialises later.  bench.py and tests/conftest.py import torch first; do the same in a process that uses this module.
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


def matrix_row_block(n_rows: int, rank: int, world: int):
    """BASELINE.json config 5 (all-frames x all-keyframes distance matrix): rank r writes rows [a, b) of the matrix into
    its own HBM; the keyframe descriptors are replicated, nothing is exchanged."""
    return rank * n_rows // world, (rank + 1) * n_rows // world


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


# ---------------------------------------------------------------------------------------------------------------
# Device-resident exchange (one rank per GPU over RCCL): top-k lists, merge, candidate hand-over and result records stay
# in HBM; the host sees ONE small copy per batch, at the end.  torch is plumbing here: device buffers, the collective
# and a handful of tiny elementwise / top-k launches on its stream; every kernel of the path itself is the library's.
def merge_topk_tensor(all_scan, k: int, base: int, n_local: int):
    """all_scan: int32 tensor (W, B, 2k + 2) = per rank and frame [k global ids (-1 padded), k counts, n_features, pad].
    Returns (win_gid (B, k) int32 global ids in rank order, -1 padded; cand_local (B, k) int32 local ids of the winners
    this rank owns, -1 elsewhere; n_feat (B,) int32).  Same order as merge_topk: count desc, global id desc."""
    import torch
    W, B, _ = all_scan.shape
    gid = all_scan[:, :, :k].permute(1, 0, 2).reshape(B, W * k).to(torch.int64)
    cnt = all_scan[:, :, k:2 * k].permute(1, 0, 2).reshape(B, W * k).to(torch.int64)
    key = torch.where(gid >= 0, (cnt << 32) | (gid + 1), torch.zeros_like(gid))       # unique per record; 0 = padding
    top = torch.topk(key, min(k, W * k), dim=1, largest=True, sorted=True).values
    if top.shape[1] < k:
        top = torch.cat([top, torch.zeros(B, k - top.shape[1], dtype=top.dtype, device=top.device)], 1)
    win_gid = torch.where(top > 0, (top & 0xFFFFFFFF) - 1, torch.full_like(top, -1)).to(torch.int32)
    mine = (win_gid >= base) & (win_gid < base + n_local)
    cand_local = torch.where(mine, win_gid - base, torch.full_like(win_gid, -1)).contiguous()
    n_feat = all_scan[:, :, 2 * k].max(dim=0).values
    return win_gid.contiguous(), cand_local, n_feat


def pick_results(res_all: np.ndarray, win_gid: np.ndarray, n_feat: np.ndarray, bases, min_features=MIN_FEATURES):
    """res_all: (W, B, 96) uint8 TickResult records of every rank; win_gid (B, k); bases: first global id of every rank.
    The anchor of a frame = most inliers, earliest position in the global candidate order on ties (M:379)."""
    W, B, _ = res_all.shape
    f64 = res_all.view(np.float64).reshape(W, B, 12)
    i32 = res_all.view(np.int32).reshape(W, B, 24)
    out = []
    for i in range(B):
        n_cand = int((win_gid[i] >= 0).sum())
        none = dict(n_inliers=0, reproj=0.0, anchor_pose=np.zeros(7), lm_idx=-1, n_candidates=n_cand)
        if n_feat[i] < min_features:
            out.append(dict(outcome=1, **{**none, "n_candidates": 0}))
            continue
        best = None
        for r in range(W):
            oc, n_inl, lm = int(i32[r, i, 18]), int(i32[r, i, 16]), int(i32[r, i, 17])
            if oc not in (0, 4) or lm < 0:
                continue
            g = lm + int(bases[r])
            pos = int(np.nonzero(win_gid[i] == g)[0][0])
            if best is None or (n_inl, -pos) > (best[0], -best[1]):
                best = (n_inl, pos, r, g, oc)
        if best is None:
            out.append(dict(outcome=3 if n_cand else 2, **none))
            continue
        n_inl, pos, r, g, oc = best
        out.append(dict(outcome=oc, n_inliers=n_inl, reproj=float(np.float32(f64[r, i, 7])), anchor_pose=f64[r, i, :7].copy(),
                        lm_idx=g, n_candidates=n_cand))
    return out


class DeviceShardedRelocalizer:
    """tick_batch of ShardedRelocalizer with the exchange on the device.  shard: HipShard with >= B slots; device: the
    torch device of this rank; group / world as for torch.distributed (world 1: no collective)."""

    def __init__(self, shard: "HipShard", rank: int, world: int, device, group=None, k: int = K_GLOBAL, bases=None):
        import torch
        self.torch, self.shard, self.rank, self.world, self.device, self.group, self.k = torch, shard, rank, world, device, group, k
        B = len(shard.engines)
        self.B = B
        self.scan_buf = torch.full((B, 2 * k + 2), -1, dtype=torch.int32, device=device)
        self.res_buf = torch.zeros((B, 96), dtype=torch.uint8, device=device)
        self.all_scan = torch.empty((world, B, 2 * k + 2), dtype=torch.int32, device=device)
        self.all_res = torch.empty((world, B, 96), dtype=torch.uint8, device=device)
        self.exch = torch.cuda.Stream(device=device)
        self.ext = [torch.cuda.ExternalStream(e.stream_ptr, device=device) for e in shard.engines]
        if bases is None:
            if world == 1:
                bases = [shard.base]
            else:
                import torch.distributed as dist
                t = torch.tensor([shard.base], dtype=torch.int64, device=device)
                allb = torch.empty(world, dtype=torch.int64, device=device)
                dist.all_gather_into_tensor(allb, t, group=group)
                bases = allb.cpu().tolist()
        self.bases = list(bases)

    def tick_batch(self, frames_dev, base_poses, seeds=None):
        torch, sh, k, B = self.torch, self.shard, self.k, len(frames_dev)
        if B > self.B:
            raise ValueError(f"batch of {B} frames on a shard with {self.B} slots")
        seeds = list(seeds) if seeds is not None else [0] * B
        es = sh.engines
        sb = self.scan_buf
        row = sb.stride(0) * 4
        if sh.n_records:
            for i in range(B):                                        # ORB + shard scan + local top-k, one stream per frame
                p = sb.data_ptr() + i * row
                es[i].tick_scan_into(frames_dev[i], sh.w, sh.h, base_poses[i], k, p, p + 4 * k, p + 8 * k)
        with torch.cuda.stream(self.exch):
            for i in range(B):
                self.exch.wait_stream(self.ext[i])
            if sh.n_records:
                ids = sb[:, :k]
                sb[:, :k] = torch.where(ids >= 0, ids + sh.base, ids)              # local -> global ids
            else:
                sb[:, :k] = -1; sb[:, k:2 * k] = 0; sb[:, 2 * k] = -1
            if self.world > 1:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self.all_scan, sb, group=self.group)  # B x (2k + 2) ints per rank
                all_scan = self.all_scan
            else:
                all_scan = sb[None]
            win_gid, cand_local, n_feat = merge_topk_tensor(all_scan[:, :B], k, sh.base, sh.n_records)
        if sh.n_records:
            for i in range(B):                                        # owners solve; a rank without a winner finishes at once
                self.ext[i].wait_stream(self.exch)
                es[i].tick_solve_from(cand_local.data_ptr() + i * k * 4, k, base_poses[i], False, seeds[i])
                es[i].d2d(self.res_buf.data_ptr() + i * 96, es[i].tick_result_dev, 96)
        with torch.cuda.stream(self.exch):
            for i in range(B):
                self.exch.wait_stream(self.ext[i])
            if not sh.n_records:
                self.res_buf.zero_()
                self.res_buf.view(torch.int32)[:, 18] = 2             # outcome no_candidates, never picked
            if self.world > 1:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self.all_res, self.res_buf, group=self.group)
                all_res = self.all_res
            else:
                all_res = self.res_buf[None]
            host = (all_res[:, :B].cpu().numpy(), win_gid.cpu().numpy(), n_feat.cpu().numpy())   # the one trip to the host
        self._keep = (cand_local, win_gid)                            # alive until the streams have passed them
        return pick_results(host[0], host[1], host[2], self.bases)
