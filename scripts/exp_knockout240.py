"""DIAGNOSTIC: what each launch of a 240-cloud call COSTS the executor (its marginal cost with eight calls in flight), next to its duration alone.
One eager call is recorded as C-ABI calls (as scripts/exp_overlap.py does); the whole list is replayed back to back on NS streams (NS calls in
flight, the executor's regime, no host work in between) and timed per call; then again with one unit left out of every replay.  Marginal cost
= (time per call with everything, measured right before and right after) - (time per call without the unit).  A unit whose marginal cost is far below its duration alone is hidden
behind the other calls' launches (dependent rounds on a few waves); one at its full duration is issue-slot-bound like its neighbours.
    python scripts/exp_knockout240.py [B=240] [precision] [NS=8]"""
import os, sys, gc, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import synthetic as syn, lbs as G, _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 240
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
CALLS = 6          # calls per stream and measurement
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}
x = torch.rand((B, 8192, 3), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
betas, pose = (torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100))
s0 = torch.cuda.Stream()
streams = [torch.cuda.Stream() for _ in range(NS)]


def step():
    model.forward_fused(x, precision=prec)
    G.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"], pose2rot=True)


rec, keep, orig = [], [], _lib.call
with torch.no_grad(), torch.cuda.stream(s0):
    step(); step()
    torch.cuda.synchronize()

    def call(name, *args):
        keep.extend(t for t in gc.get_objects() if torch.is_tensor(t) and t.is_cuda)
        rec.append((name, args))
        return orig(name, *args)
    _lib.call = call
    keep.append(step())
    _lib.call = orig
    torch.cuda.synchronize()
S0 = s0.cuda_stream
units, cur = [], None
for name, args in rec:
    if name in ("g4d_tuning_set", "g4d_tuning_set_thread"):
        continue
    if name == "g4d_launch_group_begin":
        cur = [(name, args)]
    elif cur is not None:
        cur.append((name, args))
        if name == "g4d_launch_group_end":
            units.append(("group[" + "+".join(n.replace("g4d_", "").replace("_f32", "") for n, _ in cur[1:-1]) + "]", cur))
            cur = None
    else:
        label = name.replace("g4d_", "").replace("_f32", "")
        if name == "g4d_mlp_run":
            blk = args[1].contents if hasattr(args[1], "contents") else _lib.MlpArgs.from_address(args[1])
            label = f"mlp_run(family {args[0]}, rows {blk.rows})"
        units.append((label, [(name, args)]))
_count = ctypes.c_int(0)


def replay(calls, stream):
    h = stream.cuda_stream
    for name, args in calls:
        args = [h if (isinstance(a, int) and not isinstance(a, bool) and a == S0) else a for a in args]
        if name == "g4d_launch_group_end":
            args[1] = ctypes.addressof(_count)
        orig(name, *args)


def per_call_us(skip=None):
    """NS streams x CALLS replays of the unit list (unit `skip` left out), wall time per call; median of 3."""
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_event(e0)
        ends = []
        for _ in range(CALLS):
            for st in streams:
                for i, (label, calls) in enumerate(units):
                    if i != skip:
                        replay(calls, st)
        for st in streams:
            e = torch.cuda.Event(enable_timing=True); e.record(st); ends.append(e)
        torch.cuda.synchronize()
        ts.append(max(e0.elapsed_time(e) for e in ends) * 1e3 / (NS * CALLS))
    return sorted(ts)[1]


def alone_us(calls):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0]); replay(calls, streams[0]); e1.record(streams[0])
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[2]


per_call_us()
full = per_call_us()
print(f"# {B} clouds per call, {prec}, {NS} calls in flight x {CALLS} calls per stream: {full:.0f} us per call with every launch = {B / full * 1e6:.0f} frames/s")
print(f"# {'unit':66s} {'alone':>8s} {'marginal':>9s} {'ratio':>6s}")
tot_a = tot_m = 0.0
only = os.environ.get("KNOCK", "")          # substring filter on the unit label
for i, (label, calls) in enumerate(units):
    if only and only not in label:
        continue
    a = alone_us(calls)
    f0 = per_call_us(); k = per_call_us(skip=i); f1 = per_call_us()     # the baseline on both sides of the knock-out: clocks drift over the run
    m = 0.5 * (f0 + f1) - k
    tot_a += a; tot_m += m
    print(f"  {label[:66]:66s} {a:8.1f} {m:9.1f} {m / a if a > 0 else 0:6.2f}")
print(f"# sums: alone {tot_a:.0f} us, marginal {tot_m:.0f} us (per call with everything: {full:.0f} us)")
