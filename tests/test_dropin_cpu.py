"""Host logic of the drop-in dispatch (round 6), no GPU needed: WHEN a module's forward() may go to the fused kernels, what the packers refuse,
how the twins are keyed, and the tuning fields that switch the route."""
import pytest
import torch

from garment4d_amd import fused, pointnet2_modules as PM, pytorch_utils as pt, tuning


def test_fused_route_conditions():
    sa = PM.PointnetSAModule(npoint=4, radius=0.5, nsample=4, mlp=[0, 8, 16])
    x = torch.zeros(1, 8, 3)
    with torch.no_grad():
        assert not PM.fused_route(sa.eval(), sa.mlps, x)                       # CPU tensor: the op-by-op route (whose HIP ops then refuse it loudly)
    assert not PM.fused_route(sa.eval(), sa.mlps, x.to("meta"))                # autograd on
    with torch.no_grad():
        assert not PM.fused_route(sa.train(), sa.mlps, x)
    # a tensor that LOOKS like a HIP fp32 tensor: the remaining conditions
    class Fake:
        is_cuda, dtype = True, torch.float32
    with torch.no_grad():
        assert PM.fused_route(sa.eval(), sa.mlps, Fake(), None)
        assert not PM.fused_route(sa.train(), sa.mlps, Fake())
        with PM.op_by_op():
            assert not PM.fused_route(sa.eval(), sa.mlps, Fake())
        Fake.dtype = torch.float64
        assert not PM.fused_route(sa.eval(), sa.mlps, Fake())
        Fake.dtype = torch.float32
        for bad in (pt.SharedMLP([3, 8], bn=True, preact=True), pt.SharedMLP([3, 8], bn=False, instance_norm=True),
                    pt.SharedMLP([3, 8], bn=True, activation=torch.nn.LeakyReLU(0.1))):
            assert not PM.fused_route(sa.eval(), [bad.eval()], Fake())          # stacks the kernels do not cover: op-by-op, silently
    assert not PM.fused_route(sa.eval(), sa.mlps, Fake())                      # autograd on again


def test_packers_refuse_what_the_kernels_do_not_compute():
    with pytest.raises(NotImplementedError):
        fused.pack_conv_block(pt.Conv1d(4, 4, kernel_size=3, padding=1).eval())
    with pytest.raises(NotImplementedError):
        fused.pack_conv_block(pt.Conv1d(4, 4, stride=2).eval())
    L = fused.pack_conv_block(pt.Conv1d(6, 5, bn=True).eval())
    assert (L.K, L.Cout, L.relu) == (6, 5, 1) and fused.pack_conv_block.__doc__
    blk = pt.Conv1d(6, 5, activation=None).eval()
    L1 = fused.pack_conv_block(blk)
    assert L1.relu == 0 and fused.pack_conv_block(blk) is L1                   # cached on the block ...
    with torch.no_grad():
        blk.conv.weight.mul_(2.0)
    assert fused.pack_conv_block(blk) is not L1                                # ... keyed on the parameters' version counters
    assert fused.invalidate(blk) == 1


def test_twin_is_keyed_on_the_version_counter():
    cm, pm = torch.zeros(2, 3, 5), torch.ones(2, 5, 3)
    fused.attach_twin(cm, pm)
    assert cm._g4d_pm[0] is pm and cm._g4d_pm[1] == cm._version
    cm.add_(1.0)
    assert cm._g4d_pm[1] != cm._version                                        # an in-place edit: point_major_of() will transpose again
    assert getattr(cm.clone(), "_g4d_pm", None) is None and getattr(cm[:1], "_g4d_pm", None) is None
    x = torch.zeros(1, 4, 3)
    g = (torch.zeros(4, dtype=torch.uint8), 0.1)
    assert fused.grid_of(x) is None and fused.attach_grid(x, g) is g and fused.grid_of(x) is g
    x.zero_()
    assert fused.grid_of(x) is None


def test_inference_tensors_get_no_twin():
    """torch.inference_mode() tensors keep no version counter: an in-place edit could not be noticed, so nothing is attached to them."""
    with torch.inference_mode():
        cm, pm = torch.zeros(2, 3, 5), torch.ones(2, 5, 3)
        assert fused.attach_twin(cm, pm) is cm and getattr(cm, "_g4d_pm", None) is None
        x = torch.zeros(1, 4, 3)
        fused.attach_grid(x, ("ws", 0.1))
        assert fused.grid_of(x) is None


def test_tuning_fields_of_the_dropin_route():
    t = tuning.Tuning()
    assert t.dropin_fused and t.dropin_whole_model
    t2 = t.replace(dropin_whole_model=False, native={"sa_table_dedup": 0})
    assert not t2.dropin_whole_model and dict(t2.native) == {"sa_table_dedup": 0}
    with pytest.raises(KeyError):
        t.replace(native={"dynamic_units": 1})                                 # (the counter scheduler of scripts/experiments is not a key of this library)
