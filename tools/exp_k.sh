timeout 900 python -m pytest tests/test_gpu_video.py -q -x -k "push_in_batches or stream_runner or batch_invariant" 2>&1 | tail -4
python - <<'PY'
import json, torch, bench
torch.set_grad_enabled(False)
bench.host_thread_policy()
r = bench.video_leg(torch.device("cuda:0"), precision="fp16")
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "note"}) for k, v in r.items() if k != "note"}))
PY
python bench.py --workload cfg4 --steps 40 --warmup 4 --no-cpu-baseline --clip-frames 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 clip8 default', d['value'], d['ms_per_step'])"
python bench.py --workload cfg4 --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 clip2 default', d['value'], d['ms_per_step'])"
