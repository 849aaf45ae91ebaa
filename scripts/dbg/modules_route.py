"""Per-launch table of the reference's encoder loop over the drop-in modules (Tuning.dropin_whole_model = False), 240 clouds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garment4d_amd import _lib, tuning
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((240, 8192, 3), generator=g, device="cuda")
for whole in (True, False):
    with torch.no_grad(), tuning.use(tuning.current().replace(dropin_whole_model=whole)):
        model(x); model(x); torch.cuda.synchronize()
        with _lib.timed_calls() as t:
            model(x)
        res = t.results()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); model(x); e1.record(); torch.cuda.synchronize()
    print(f"## dropin_whole_model={whole}: {len(res)} C-ABI calls, {sum(r[2] for r in res):.0f} us in them, {e0.elapsed_time(e1)*1e3:.0f} us end to end")
    for name, ints, us in res:
        print(f"  {name:40s} {us:8.1f}")
