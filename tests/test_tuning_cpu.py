"""garment4d_amd/tuning.py: the kernel-selection state as one explicit immutable object (CPU: no launches)."""
import dataclasses
import threading

import pytest

from garment4d_amd import tuning


def test_tuning_is_immutable_and_replace_validates_native_keys():
    t = tuning.current()
    with pytest.raises(dataclasses.FrozenInstanceError):
        t.fp_cells = False
    u = t.replace(fp_cells=False, native={"sa_table_min_rows": 0})
    assert u.fp_cells is False and t.fp_cells is True and dict(u.native) == {"sa_table_min_rows": 0}
    assert dict(u.replace(native={"gemm_tile": 0}).native) == {"sa_table_min_rows": 0, "gemm_tile": 0}      # a mapping merges
    assert u.replace(native=()).native == ()                                                                 # a tuple replaces
    with pytest.raises(KeyError):
        t.replace(native={"no_such_key": 1})
    with pytest.raises(TypeError):
        t.replace(no_such_field=1)


def test_use_nests_and_is_per_thread():
    base = tuning.current()
    seen = {}
    with tuning.use(base.replace(grid_min_n=1)):
        assert tuning.current().grid_min_n == 1
        with tuning.use(tuning.current().replace(fp_table=False)):
            assert tuning.current().grid_min_n == 1 and tuning.current().fp_table is False
        assert tuning.current().fp_table is True

        def other():                      # a thread started inside the block still sees the process default, not this thread's override
            seen["other"] = tuning.current().grid_min_n
        th = threading.Thread(target=other)
        th.start(); th.join()
    assert tuning.current() is base and seen["other"] == tuning.DEFAULT.grid_min_n


def test_legacy_module_names_are_read_only_views():
    from garment4d_amd import fused
    assert fused.FP_CELLS == tuning.current().fp_cells and fused.GRID_MIN_N == tuning.current().grid_min_n
    with tuning.use(tuning.current().replace(fp_cells=False)):
        assert fused.FP_CELLS is False
    with pytest.raises(AttributeError):
        fused.NO_SUCH_SWITCH
