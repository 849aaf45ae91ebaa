"""DIAGNOSTIC ONLY (not a valid throughput number): what each launch of the cfg2 step costs the 16-stream mix.  bench.py is run with chosen
C-ABI calls skipped (their outputs keep whatever the eager warm-up left there); the drop in ms_per_step is that launch's marginal cost.
Only launches whose outputs are not used as indices are knocked out (MLP stacks, tables, lbs).
    KNOCK="g4d_linear_f32#0,g4d_mlp_chain_group_table_ws_f32#0" python scripts/exp_knockout.py [bench.py flags]
'#k' = the k-th call of that entry point within a step (a step starts at g4d_fps_gather_grid_f32); without '#k' every call; '*' globs names.
G4D_KNOCK_TRACE=1 prints the C-ABI calls of one step."""
import fnmatch, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch  # noqa: F401
from garment4d_amd import _lib

spec = [s for s in os.environ.get("KNOCK", "").split(",") if s]
knock = set()
for s in spec:
    name, _, k = s.partition("#")
    knock.add((name, int(k) if k else None))
orig = _lib.call
count = {}
warm = {"steps": 0}
STEP_HEAD = "g4d_fps_gather_grid_f32"


def call(name, *args):
    if name == STEP_HEAD:
        count.clear()
        warm["steps"] += 1
    k = count.get(name, 0)
    count[name] = k + 1
    if warm["steps"] == 2 and os.environ.get("G4D_KNOCK_TRACE"):
        print("call", name, "#%d" % k, file=sys.stderr)
    if warm["steps"] > 2 and ((name, k) in knock or (name, None) in knock or any(fnmatch.fnmatch(name, pat) for pat, kk in knock if "*" in pat and kk in (None, k))):   # the first eager steps run everything (valid outputs / indices everywhere)
        return 0
    return orig(name, *args)


_lib.call = call
sys.argv = ["bench.py", "--no-cpu-baseline", "--warmup", "4"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
