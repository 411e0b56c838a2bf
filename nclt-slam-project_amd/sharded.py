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

Load order: PyTorch-ROCm ships its own HIP runtime, and a process must run on ONE.  `_native.load()` sees to that whatever
the import order (it maps torch's runtime first when torch is installed but not imported yet):
nothing has to be imported first.
"""
from __future__ import annotations

import numpy as np

from .landmarks import shard_by_rows

K_GLOBAL = 25
MIN_FEATURES = 10      # MIN_MATCHES default: fewer current keypoints -> curr_no_features (M:307); engines carry their own


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
        # curr_no_features gate (M:307): the matcher parameter of the shard's engine when there is one
        eng = getattr(backend, "engine", None)
        self.min_features = int(eng.get_params().min_matches) if eng is not None and hasattr(eng, "get_params") else MIN_FEATURES

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
            packed[i, 2 * k] = scan[2] if len(scan) > 2 else self.min_features
        allp = self._all_gather(packed)                                       # (world, B, 2k + 1)
        n_feat = allp[:, :, 2 * k].max(axis=0)
        winners, jobs = [], []
        for i in range(B):
            if n_feat[i] < self.min_features:
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
            res[i, 3] = 1 if n_feat[i] < self.min_features else (3 if len(winners[i][0]) else 2)
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


class _Batch:
    """one batch in flight on a group's stream (DeviceShardedRelocalizer.submit)"""
    __slots__ = ("group", "n", "done")

    def __init__(self, group, n):
        self.group, self.n, self.done = group, n, False


class DeviceShardedRelocalizer:
    """tick_batch of ShardedRelocalizer with the exchange on the device and several batches in flight.

    A *group* = `batch` contexts that share ONE stream and the rank's shard (reloc_db_share), plus the group's exchange
    buffers.  A batch is three library calls and two collectives, all enqueued on the group's stream, nothing waited for:
        reloc_shard_scan_batch_dev   ORB per frame, ONE scan launch for the batch, per-frame ranking -> rows of global ids
        all-gather                   B x (2k + 2) int32 per rank (RCCL over xGMI; nothing at world 1)
        reloc_shard_merge_dev        identical merge on every rank -> winners, the ones this rank owns
        reloc_shard_solve_batch_dev  owners solve; result records stored straight into the gather buffer
        all-gather                   B x 96 bytes per rank
        one copy of (results, winners, feature counts) into pinned host memory, then an event
    `depth` groups take batches round-robin, so the scan of batch i + 1 is on the device before the exchange of batch i
    has finished (round 2 ran one batch at a time with three host synchronisations in it and per-frame launches: half
    the unsharded rate at equal records per rank).  submit() enqueues, result() waits for that batch's event only.

    group: None -> torch.distributed default group; an object with all_gather_tensor(rank, out, inp, stream) stands in for
    it when the ranks are threads of one process (tests)."""

    def __init__(self, shard: "HipShard", rank: int, world: int, device, group=None, k: int = K_GLOBAL, bases=None,
                 batch: int | None = None, depth: int = 2, n_streams: int = 4, force_collective: bool = False):
        """force_collective: route both exchanges through the collective at world size 1 too (an initialised process group of
        one rank, or an injected group) instead of the single-rank device copy -- the way to run the RCCL transport on a
        one-GPU box (tests, bench.py --shard-db --force-dist)."""
        import torch
        from .engine import Engine
        self.torch, self.shard, self.rank, self.world, self.device, self.group, self.k = torch, shard, rank, world, device, group, k
        self.collective = world > 1 or bool(force_collective)
        B = batch if batch is not None else max(1, min(8, len(shard.engines)))
        self.B, self.depth = B, depth
        e0 = shard.engine
        self.min_features = int(e0.get_params().min_matches) if shard.n_records else MIN_FEATURES
        row = 2 * k + 2
        self.groups = []
        # the groups' streams are made first and one after the other: the HIP runtime hands a new stream the least-used of
        # its (four) hardware queues, and two groups whose streams share a hardware queue run strictly one after the other
        # (four hardware queues: more than four streams only share them; further groups take turns on the four streams, which
        # is what lets the host run `depth` batches ahead of the device)
        streams = [torch.cuda.Stream(device=device) for _ in range(min(depth, n_streams))]
        for g in range(depth):
            ts = streams[g % len(streams)]
            engines = []
            if shard.n_records:
                for _ in range(B):
                    e = Engine(e0.device, e0.max_w, e0.max_h, e0.max_feat)
                    e.db_share(e0)
                    e.set_params_from(e0)
                    e.set_stream(ts.cuda_stream)
                    engines.append(e)
            # what the host reads of a batch -- every rank's result records, the winners, the feature counts -- is ONE
            # contiguous device buffer (the tensors below are views of it) and ONE copy into pinned memory per batch
            n_res, n_win, n_nf = world * B * 96, B * k * 4, B * 4
            pack = torch.zeros(n_res + n_win + n_nf, dtype=torch.uint8, device=device)
            h_pack = torch.empty(n_res + n_win + n_nf, dtype=torch.uint8).pin_memory()
            all_res = pack[:n_res].view(world, B, 96)
            self.groups.append(dict(
                stream=ts, engines=engines,
                scan=torch.full((B, row), -1, dtype=torch.int32, device=device),
                all_scan=torch.empty((world, B, row), dtype=torch.int32, device=device),
                win_gid=pack[n_res:n_res + n_win].view(torch.int32).view(B, k),
                cand_local=torch.empty((B, k), dtype=torch.int32, device=device),
                n_feat=pack[n_res + n_win:].view(torch.int32),
                all_res=all_res, pack=pack, h_pack=h_pack,
                # the rank's own result records when a collective moves them (its send buffer, made once: no allocation on the
                # group's stream per batch); without a collective they are written straight into all_res[rank]
                send_res=torch.zeros((B, 96), dtype=torch.uint8, device=device) if self.collective else None,
                h_res=h_pack[:n_res].view(world, B, 96),
                h_win=h_pack[n_res:n_res + n_win].view(torch.int32).view(B, k),
                h_nfeat=h_pack[n_res + n_win:].view(torch.int32),
                event=torch.cuda.Event(), pending=None))
        self._next = 0
        if bases is None:
            if not self.collective:
                bases = [shard.base]
            else:
                t = torch.tensor([shard.base], dtype=torch.int64, device=device)
                allb = torch.empty(world, dtype=torch.int64, device=device)
                self._all_gather(allb, t, torch.cuda.current_stream(device))
                bases = allb.cpu().tolist()
        self.bases = list(bases)

    def _all_gather(self, out, inp, stream):
        """out (world, ...) <- inp of every rank, ordered on `stream`"""
        if not self.collective:
            out[0].copy_(inp)
        elif hasattr(self.group, "all_gather_tensor"):
            self.group.all_gather_tensor(self.rank, out, inp, stream)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(out, inp, group=self.group)    # stream-ordered against the current stream

    def submit(self, frames_dev, base_poses, seeds=None) -> _Batch:
        """enqueue one batch (<= `batch` frames) on the next group's stream; nothing is waited for unless that group's
        previous batch has not been collected yet"""
        torch, sh, k, n = self.torch, self.shard, self.k, len(frames_dev)
        if n > self.B or n < 1:
            raise ValueError(f"batch of {n} frames on groups of {self.B} slots")
        g = self.groups[self._next]
        self._next = (self._next + 1) % self.depth
        if g["pending"] is not None and not g["pending"].done:
            g["event"].synchronize()                              # its buffers are still in use
            g["pending"].done = True
        from .engine import Engine
        seeds = list(seeds) if seeds is not None else [0] * n
        row = 2 * k + 2
        with torch.cuda.stream(g["stream"]):
            if sh.n_records:
                Engine.shard_scan_batch_dev(g["engines"][:n], frames_dev, sh.w, sh.h, base_poses, k, sh.base, g["scan"].data_ptr())
            else:                                                 # more ranks than records: this rank only takes part in the exchange
                g["scan"][:, :k] = -1; g["scan"][:, k:2 * k] = 0; g["scan"][:, 2 * k] = -1
            if self.collective:
                self._all_gather(g["all_scan"], g["scan"], g["stream"])
                all_scan, stride = g["all_scan"], self.B * row
            else:
                all_scan, stride = g["scan"], self.B * row
            eng = g["engines"][0] if sh.n_records else None
            if eng is not None:
                eng.shard_merge_dev(all_scan.data_ptr(), self.world, stride, n, k, sh.base, sh.n_records, g["win_gid"].data_ptr(),
                                    g["cand_local"].data_ptr(), g["n_feat"].data_ptr())
                res = g["send_res"] if self.collective else g["all_res"][self.rank]
                Engine.shard_solve_batch_dev(g["engines"][:n], g["cand_local"].data_ptr(), k, base_poses, seeds, res.data_ptr())
            else:
                wg, _, nf = merge_topk_tensor(all_scan.view(self.world, self.B, row)[:, :n], k, sh.base, 0)
                g["win_gid"][:n] = wg; g["n_feat"][:n] = nf
                res = g["send_res"] if self.collective else g["all_res"][self.rank]
                res.zero_()
                res.view(torch.int32)[:, 18] = 2                  # outcome no_candidates, never picked
            if self.collective:
                self._all_gather(g["all_res"], g["send_res"], g["stream"])
            g["h_pack"].copy_(g["pack"], non_blocking=True)                      # the one trip to the host, not waited for here
            g["event"].record(g["stream"])
        b = _Batch(g, n)
        g["pending"] = b
        return b

    def result(self, b: _Batch):
        """wait for that batch (its event only) and pick every frame's anchor"""
        g = b.group
        if g["pending"] is not b:
            raise RuntimeError("this batch's buffers have been reused: collect a batch before `depth` more are submitted")
        g["event"].synchronize()
        b.done = True
        n = b.n
        return pick_results(np.ascontiguousarray(g["h_res"].numpy()[:self.world, :n]), g["h_win"].numpy()[:n].copy(),
                            g["h_nfeat"].numpy()[:n].copy(), self.bases, self.min_features)

    def tick_batch(self, frames_dev, base_poses, seeds=None):
        return self.result(self.submit(frames_dev, base_poses, seeds))

    def close(self):
        for g in self.groups:
            g["event"].synchronize() if g["pending"] is not None else None
            g["stream"].synchronize()
            for e in g["engines"]:
                e.close()
            g["engines"] = []
