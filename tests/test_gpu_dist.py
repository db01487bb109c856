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
