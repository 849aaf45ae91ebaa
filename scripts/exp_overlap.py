"""DIAGNOSTIC: which launches of the cfg2 step run CONCURRENTLY with which -- the question behind "the executor's calls in flight overlap
so little" (the knock-out table of a coalesced call is additive).  One eager step on B clouds is recorded as C-ABI calls (entry point +
arguments; a launch group is one unit), then units are replayed from their recorded arguments: alone, and in pairs on two streams started
together.  Columns: t(a) alone, t(b) alone, both together (a launched first), together / (t(a) + t(b)), together / max.
    python scripts/exp_overlap.py [B=240] [precision]            env PAIRS="fps_gather_grid:sa_xyz,..." to choose the pairs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import synthetic as syn, lbs as G, _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 240
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand((B, 8192, 3), generator=g, device=dev)
betas, pose = (torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100))
s0, sa, sb = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def step():
    model.forward_fused(x, precision=prec)
    G.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"], pose2rot=True)


rec = []
orig = _lib.call
keep = []          # everything a recorded call points at stays alive
with torch.no_grad(), torch.cuda.stream(s0):
    step(); step()
    torch.cuda.synchronize()

    import gc

    def call(name, *args):   # every device tensor alive at the time of a call is kept: no block of this step is handed to a later tensor,
        keep.extend(t for t in gc.get_objects() if torch.is_tensor(t) and t.is_cuda)   # so every unit can be replayed on valid inputs
        rec.append((name, args))
        return orig(name, *args)
    _lib.call = call
    out = step()
    keep.append(out)
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
        units.append((name.replace("g4d_", "").replace("_f32", ""), [(name, args)]))
seen = {}
named = []
for label, calls in units:
    k = seen.get(label, 0)
    seen[label] = k + 1
    named.append((f"{label}#{k}", calls))


import ctypes
_count = ctypes.c_int(0)


def replay(calls, stream):
    h = stream.cuda_stream
    for name, args in calls:
        args = [h if (isinstance(a, int) and not isinstance(a, bool) and a == S0) else a for a in args]
        if name == "g4d_launch_group_end":   # its second argument pointed at a local of the recorded call
            args[1] = ctypes.addressof(_count)
        orig(name, *args)


def timed(fn, reps=5):
    best = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        sb.wait_event(e0)
        fn()
        e1.record(sa); e2.record(sb)
        torch.cuda.synchronize()
        best.append((max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e2) * 1e3))
    best.sort()
    global last_ends
    last_ends = best[len(best) // 2][1:]          # when stream a / stream b finished in the median run
    return best[len(best) // 2][0]


last_ends = (0.0, 0.0)


alone = {}
print(f"# {B} clouds per call, {prec}; us, median of 5")
for label, calls in named:
    alone[label] = timed(lambda: replay(calls, sa))
    print(f"alone {label:70s} {alone[label]:9.1f}")
print(f"sum of units {sum(alone.values()):9.1f}")

pairs = os.environ.get("PAIRS", "")
if pairs:
    want = [p.split(":") for p in pairs.split(",")]
else:
    heavy = [l for l, _ in named if alone[l] > 100.0]
    first = [l for l in heavy if l.startswith("fps_gather_grid")]
    want = [(a, b) for a in first for b in heavy if b != a]
    nn = [l for l in heavy if l.startswith("three_nn")]
    want += [(a, b) for a in nn for b in heavy if b != a and not b.startswith("fps")]
byname = dict(named)


def find(pat):
    hits = [l for l in byname if l.startswith(pat)]
    return hits[0] if hits else None


print("# pairs: a | b | t(a) | t(b) | a then b started together | / sum | / max || b then a")
for a, b in want:
    a, b = find(a), find(b)
    if a is None or b is None:
        continue
    t_ab = timed(lambda: (replay(byname[a], sa), replay(byname[b], sb)))
    ends = last_ends
    t_ba = timed(lambda: (replay(byname[b], sb), replay(byname[a], sa)))
    ta, tb = alone[a], alone[b]
    print(f"{a[:44]:44s} | {b[:44]:44s} | {ta:7.1f} | {tb:7.1f} | {t_ab:7.1f} | {t_ab / (ta + tb):5.2f} | {t_ab / max(ta, tb):5.2f} || {t_ba:7.1f} || a ended {ends[0]:7.1f}, b ended {ends[1]:7.1f}")
