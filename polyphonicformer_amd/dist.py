"""Multi-GPU side of the hot path: one process per GPU, `torch.distributed`
("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

* Frames are independent in a1-a7, so they are SHARDED across ranks with no data-path collective
  (`shard_frames`).  Weights (~50 MB) are replicated.
* The only exchange is for video: the tracker (`QuasiDenseEmbedTracker.match`,
  polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:137-207) is stateful and strictly
  sequential in frame order, but consumes only per-frame records (bboxes[n,5], labels[n],
  embeds[n,256]) with n <= max_per_img.  Each rank pads its frames' records to a fixed
  [frames, max_per_img, 262] fp32 block (~105 KB/frame) and ONE all-gather per step hands every rank
  all records, which are then replayed in frame order -- bit-identical to the single-process order.
  The payload is latency bound (far below the ~153 GB/s of one xGMI link), so a single batched
  collective is the right shape; feature maps are never exchanged."""
import torch
import torch.distributed as dist

REC_COLS = 5 + 1 + 256   # bbox(5) | label | embed(256)


def shard_frames(num_frames, rank, world):
    """contiguous chunk of frame indices owned by `rank` (videos keep temporal locality per rank)"""
    base, rem = divmod(num_frames, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def pack_track_records(bboxes, labels, embeds, max_per_img=100):
    """one frame -> ([max_per_img, 262] fp32 block, count)"""
    n = int(bboxes.shape[0])
    if n > max_per_img:
        raise ValueError(f"{n} detections > max_per_img={max_per_img}")
    rec = torch.zeros((max_per_img, REC_COLS), dtype=torch.float32, device=bboxes.device)
    rec[:n, :5] = bboxes.float()
    rec[:n, 5] = labels.float()          # class ids < 2^24: exact in fp32
    rec[:n, 6:] = embeds.float()
    return rec, n


def unpack_track_records(rec, n):
    return rec[:n, :5], rec[:n, 5].long(), rec[:n, 6:]


def allgather_track_records(frame_ids, records, counts, frames_per_rank, max_per_img=100):
    """frame_ids: list[int] global frame indices of this rank's frames (len <= frames_per_rank);
    records: list of [max_per_img, 262] blocks; counts: list[int].
    Returns the list of (frame_id, bboxes, labels, embeds) of ALL ranks sorted by frame id."""
    dev = records[0].device if records else torch.device("cpu")
    world = dist.get_world_size() if dist.is_initialized() else 1
    blk = torch.zeros((frames_per_rank, max_per_img, REC_COLS), dtype=torch.float32, device=dev)
    meta = torch.full((frames_per_rank, 2), -1, dtype=torch.int64, device=dev)
    for i, (fid, rec, n) in enumerate(zip(frame_ids, records, counts)):
        blk[i] = rec
        meta[i, 0], meta[i, 1] = fid, n
    if world > 1:
        # concatenated along dim 0 (the layout every backend accepts), viewed per rank below
        all_blk = torch.empty((world * frames_per_rank,) + tuple(blk.shape[1:]), dtype=blk.dtype, device=dev)
        all_meta = torch.empty((world * frames_per_rank, 2), dtype=meta.dtype, device=dev)
        dist.all_gather_into_tensor(all_blk, blk)
        dist.all_gather_into_tensor(all_meta, meta)
        all_blk = all_blk.view((world, frames_per_rank) + tuple(blk.shape[1:]))
        all_meta = all_meta.view(world, frames_per_rank, 2)
    else:
        all_blk, all_meta = blk[None], meta[None]
    out = []
    am = all_meta.cpu()
    for r in range(all_blk.shape[0]):
        for i in range(frames_per_rank):
            fid, n = int(am[r, i, 0]), int(am[r, i, 1])
            if fid >= 0:
                out.append((fid,) + tuple(unpack_track_records(all_blk[r, i], n)))
    out.sort(key=lambda t: t[0])
    return out


def barrier_and_max(value, device):
    """bench contract: barrier, then the MAX over ranks of a python float"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- data-parallel training (the reference wraps the detector in MMDistributedDataParallel, tools/train.py) ------
def reduce_mean(t, group=None):
    """mmdet.core.utils.reduce_mean (kernel_update_head.py:376-377): the mean over the ranks of `group` (None: the default
    group) of a scalar tensor; identity on one"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size(group)), op=dist.ReduceOp.SUM, group=group)
    return t


class GradBuckets:
    """Bucketed gradient all-reduce overlapped with backward, sized for xGMI rings rather than NVSwitch: a ring all-reduce
    moves 2 (W - 1) / W of the payload over each GPU's slowest link, so few large buckets (default 32 MiB: the path's 54 MB of
    fp32 gradients leave in two collectives) beat many small ones.  Parameters are bucketed in REVERSE registration order
    (the order autograd finishes them in); a post-accumulate hook on each parameter counts its bucket down and the last one
    flattens the bucket and starts an asynchronous all-reduce on it while backward carries on.  `finish()` waits, divides
    by the world size and scatters the means back into `.grad`.  With one rank nothing is sent."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._left, self._work, self._flat = [], [], []
        self._armed = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in self.params]

    def start(self):
        """arm the buckets for one backward pass (call before every `backward`); a backward outside start() .. finish() --
        another TrainStep on the same heads, a fine-tuning script -- leaves the buckets alone"""
        self._left = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._armed = True

    def _ready(self, p):
        if not self._armed:
            return
        i = self._of[id(p)]
        assert self._left[i] > 0, "a parameter's gradient arrived twice in one armed backward pass"
        self._left[i] -= 1
        if self._left[i] == 0:
            self._launch(i)

    def _launch(self, i):
        if self.world == 1:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in self.buckets[i]])
        self._flat[i] = flat
        self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """wait for every bucket; parameters whose gradient never arrived (unused this step) are reduced as zeros"""
        if self.world == 1:
            self._armed = False
            return
        for i, left in enumerate(self._left):
            if left > 0:                      # some parameter of the bucket took no part in this backward
                self._left[i] = 0
                self._launch(i)
        for i, b in enumerate(self.buckets):
            self._work[i].wait()
            flat, o = self._flat[i].div_(self.world), 0
            for p in b:
                n = p.numel()
                g = flat[o:o + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                o += n
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
