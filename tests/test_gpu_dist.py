"""GPU: the one collective of the path on the backend it ships on -- `dist.allgather_track_records` over "nccl"
(= RCCL on ROCm) with world size = the number of visible GPUs (1 on the single-GPU test box, 8 on a full node), one
process per GPU, payload checked bit for bit; plus `bench.py --gpus 2` launching its own ranks (on one GPU the two
ranks share the device over gloo: the launch / rendezvous / max-over-ranks path, not a throughput number)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(fid):
    g = torch.Generator().manual_seed(5000 + fid)
    n = int(torch.randint(1, 100, (1,), generator=g))
    return torch.rand(n, 5, generator=g), torch.randint(0, 8, (n,), generator=g), torch.randn(n, 256, generator=g)


def _worker(rank, world, port, per, q):
    import torch.distributed as dist
    from polyphonicformer_amd import dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    mine = D.shard_frames(world * per, rank, world)
    recs, cnts = [], []
    for f in mine:
        r, n = D.pack_track_records(*[t.to(dev) for t in _records(f)])
        recs.append(r)
        cnts.append(n)
    out = D.allgather_track_records(mine, recs, cnts, per)
    ok = [t[0] for t in out] == list(range(world * per))
    for fid, bb, lab, emb in out:
        b0, l0, e0 = _records(fid)
        ok &= bb.is_cuda and torch.equal(bb.cpu(), b0) and torch.equal(lab.cpu(), l0) and torch.equal(emb.cpu(), e0)
    mx = D.barrier_and_max(float(rank + 1), dev)
    q.put((rank, bool(ok), mx))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_track_records_on_rccl(gpu):
    world = torch.cuda.device_count()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, 3, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert all(r[2] == float(world) for r in res)


def test_bench_launches_its_own_ranks(gpu):
    """plain `python bench.py --gpus 2` (no outer launcher): rank 0 starts rank 1 itself, ONE JSON line comes back with
    n_gpus 2, the aggregate of both ranks and the timed track-record all-gather.  With fewer than two GPUs the two
    ranks share GPU 0 and rendezvous over gloo (PH_DIST_BACKEND), which exercises the same launch path."""
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["PH_DIST_BACKEND"] = "gloo"
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny", "--frames", "8",
                        "--steps", "3", "--warmup", "1", "--streams", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    tag = res["track_allgather"]
    assert tag["2"]["world_size"] == 2 and tag["2"]["payload_round_trip_exact"] and tag["2"]["collective_us_per_step"] > 0


# ---- cfg4: clips sharded over the ranks, ids equal to the single-process video loop ------------------------------------------
def _cfg4_worker(rank, world, port, backend, steps, q, warmup=0):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    # the bench's host thread policy (bench.host_thread_policy; here scaled to the ranks that share this host): torch's default intra-op
    # pool is one spinning thread per hardware thread PER PROCESS -- eight ranks of them turned this test's host work (merge accept loops,
    # the tracker's CPU ops) into 12 minutes
    torch.set_num_threads(max(1, min(bench.HOST_THREADS, 32 // world)))
    run, _ = bench.cfg4_run(dev, world, rank, backend, precision="fp16", clip_frames=2, steps=steps, warmup=warmup, collect_ids=True)
    q.put((rank, run))
    dist.barrier()
    dist.destroy_process_group()


def _cfg4(world, backend, steps, warmup=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cfg4_worker, args=(r, world, port, backend, steps, q, warmup)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=900) for _ in ps)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _same_ids(runs, one, world, shared_gpu):
    """every rank's ids equal the single-process loop's.  When the ranks SHARE one GPU (a one-GPU box, gloo) the persistent one-pass
    KernelHead launches of different processes can starve each other past the hand-off bound; the calls that gave up were redone by the
    predicated two-pass kernels inside the same call (valid results, ~1e-6 from the one-pass ones, so a detection score next to a
    threshold may fall on the other side): such a run says so (`khead_onepass_timeouts`) and is not held to bit identity -- what it
    still must show is the sharded path itself: every frame present on every rank, all ranks agreeing with each other"""
    gave_up = sum(int(runs[r].get("khead_onepass_timeouts") or 0) for r in range(world)) + int(one.get("khead_onepass_timeouts") or 0)
    for r in range(world):
        assert runs[r]["world_size"] == world
        assert sorted(runs[r]["track_ids"]) == sorted(one["track_ids"]) and runs[r]["track_ids"] == runs[0]["track_ids"], r
        if not (shared_gpu and gave_up):
            assert runs[r]["track_ids"] == one["track_ids"], (r, gave_up, runs[r]["track_ids"], one["track_ids"])
    if shared_gpu and gave_up:
        print(f"world {world} on one shared GPU: {gave_up} one-pass KernelHead time-outs (two-pass results inside those calls); bit identity not asserted")


def test_cfg4_sharded_clips_track_ids_equal_the_single_process_video_loop(gpu):
    """BASELINE configs[3] (`bench.py --workload cfg4`): the video's frames sharded as one clip per rank, ONE all-gather of
    the track records per step, the tracker replayed in frame order -- the integer track ids of every frame must equal,
    bit for bit, what ONE process gets that walks the same frames in order (polyphonic/apis/video_inference.py:8-31).
    Ranks = the visible GPUs over nccl (= RCCL); on a single-GPU box additionally two ranks sharing the GPU over gloo, so
    that the sharded path (shard_frames, all-gather, replay with a persistent tracker) is crossed with world > 1."""
    ndev = torch.cuda.device_count()
    # the same 8 frames: world 1 walks them as 4 steps of one 2-frame clip; world 2 as 2 steps of two clips; world = ndev ...
    one = _cfg4(1, "nccl", 4)[0]
    assert one["world_size"] == 1 and len(one["track_ids"]) == 8
    assert sum(len(v) for v in one["track_ids"].values()) > 0, "no thing segment was tracked: the comparison would be vacuous"
    two = _cfg4(2, "nccl" if ndev >= 2 else "gloo", 2)
    _same_ids(two, one, 2, shared_gpu=ndev < 2)
    if ndev >= 4 and 8 % (2 * ndev) == 0:
        allr = _cfg4(ndev, "nccl", 8 // (2 * ndev))
        for r in range(ndev):
            assert allr[r]["track_ids"] == one["track_ids"]


def test_bench_cfg4_json_line(gpu):
    """`python bench.py --workload cfg4 --gpus N`: one JSON line with the metric, the collective's time, roofline and
    cpu_baseline; the process group spans exactly --gpus ranks"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "cfg4", "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["unit"] == "frames/s" and res["scaling"] == "weak"
    assert res["cfg4"]["world_size"] == 1 and res["cfg4"]["allgather_track_records_us_per_step"] > 0
    assert res["roofline"]["kernel"] == "pool" and 0 < res["roofline"]["frac"] < 1


# ---- the 8-rank shape, turnkey: on a box with fewer than 8 GPUs the ranks share GPU 0 and rendezvous over gloo ----
def _bench_line(args, env_extra=None, timeout=1500):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus8_self_launch_json_line(gpu):
    """`python bench.py --gpus 8` exactly as the driver's 8-GPU run starts it (minus the outer launcher): rank 0 starts ranks 1..7,
    every rank times its own frames, the line carries n_gpus 8, the max-over-ranks step time, the 8-rank all-gather (payload of all
    8 ranks bit-exact) and value = frames of all ranks / that time"""
    extra = {} if torch.cuda.device_count() >= 8 else {"PH_DIST_BACKEND": "gloo"}
    res = _bench_line(["--gpus", "8", "--workload", "tiny", "--frames", "8", "--steps", "3", "--warmup", "1", "--streams", "2", "--no-cpu-baseline",
                       "--no-kernel-head"], extra)
    assert res["n_gpus"] == 8 and res["scaling"] == "weak" and res["steps"] == 3 and res["value"] > 0
    assert abs(res["value"] - 8 * 8 * 1e3 / res["ms_per_step"]) / res["value"] < 1e-3          # whole-job aggregate over the 8 ranks
    for fpr in ("2", "8"):
        tag = res["track_allgather"][fpr]
        assert tag["world_size"] == 8 and tag["payload_round_trip_exact"] and tag["collective_us_per_step"] > 0
    assert res["roofline"]["frac"] > 0 and "8 GPU(s)" in res["config"]["parallelism"]


def test_cfg4_world8_ids_equal_world1(gpu):
    """BASELINE configs[3] at its 8-rank shape: 16 frames as 8 two-frame clips, one per rank, ONE all-gather, the replay on every rank:
    the track ids of all 16 frames equal the single-process loop's on all 8 ranks.  (The timed steps follow the warm-up and calibration
    steps: world 8 with one step times frames 16 .. 31, world 1 reaches them with 2 warm-up + 6 calibration steps of 2 frames.)"""
    ndev = torch.cuda.device_count()
    backend = "nccl" if ndev >= 8 else "gloo"
    one = _cfg4(1, "nccl", 8, warmup=2)[0]
    assert sorted(one["track_ids"]) == list(range(16, 32)) and sum(len(v) for v in one["track_ids"].values()) > 0
    eight = _cfg4(8, backend, 1)
    _same_ids(eight, one, 8, shared_gpu=ndev < 8)


# ---- data-parallel training step: two ranks, gradients averaged by dist.GradBuckets -------------------------------------------
def _train_heads(dev):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers as Hh
    from test_gpu_loss import _rpn_head
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_update  # noqa: F401
    rpn, sd = _rpn_head(dev)
    roi_a = dict(type='MaskHungarianAssignerWithDepth', cls_cost=dict(type='FocalLossCost', weight=2.0),
                 dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True), mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    roi = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=2, assign_stages=2, stage_loss_weights=[1] * 2, num_proposals=100,
                           num_thing_classes=8, num_stuff_classes=11, mask_head=Hh.stage_cfg(256, 2048, 8, 19, 8, 11),
                           train_cfg=dict(assigner=roi_a, sampler=dict(type='MaskPseudoSampler'), pos_weight=1.)))
    roi.load_state_dict({k[len("roi_head."):]: v for k, v in sd.items() if k.startswith("roi_head.") and int(k.split(".")[2]) < 2})
    return rpn, roi.to(dev)


def _train_batch(rank, dev):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers as Hh
    B, H, W = 1, 8, 16
    feats = [f.to(dev) for f in Hh.neck_inputs(300 + rank, B, 256, H, W)]
    gts = [{k: v.to(dev) for k, v in g.items()} for g in Hh.train_gt(400 + rank, B, 2 * H, 2 * W, 8, 11, [3 + rank])]
    gd = torch.stack([g["depth"][None] for g in gts])
    return (feats, [Hh.img_meta(H * 8, W * 8)] * B, [g["masks"] for g in gts], [g["labels"] for g in gts], [g["sem_seg"] for g in gts],
            [g["sem_cls"] for g in gts], gd)


def _ddp_worker(rank, world, port, backend, q):
    import torch.distributed as dist
    from polyphonicformer_amd import train as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    rpn, roi = _train_heads(dev)
    step = T.TrainStep(rpn, roi, bucket_bytes=8 << 20)
    losses, total, _ = step.forward_backward(*_train_batch(rank, dev))
    names = [n for n, _ in rpn.named_parameters()] + [n for n, _ in roi.named_parameters()]
    digest = {n: (float(p.grad.double().norm()), float(p.grad.double().sum())) for n, p in zip(names, step.parameters())}
    q.put((rank, len(step.buckets.buckets), float(total), digest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_averages_gradients(gpu):
    """two ranks (two GPUs over RCCL when there are two, else both on GPU 0 over gloo), each with its own image and its own
    objective: after `TrainStep.forward_backward` (bucketed all-reduce started from the gradient hooks during backward,
    focal normaliser = mean positive count over the ranks, kernel_update_head.py:376) both ranks hold the SAME gradient for
    every one of the parameter tensors.  (That the reduced value is the mean of the ranks' gradients: tests/test_dist_gloo.py.)"""
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=600) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] >= 2                                  # several buckets
    assert abs(res[0][2] - res[1][2]) > 1e-3               # different images, different objectives
    d0, d1 = res[0][3], res[1][3]
    assert len(d0) > 150                                   # every parameter tensor of the two heads (two stages here)
    for n in d0:                                           # both ranks hold the same averaged gradients
        assert abs(d0[n][0] - d1[n][0]) <= 1e-6 * max(1.0, d0[n][0]), n
        assert abs(d0[n][1] - d1[n][1]) <= 1e-5 * max(1.0, d0[n][0]), n
