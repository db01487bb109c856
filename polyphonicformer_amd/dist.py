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
