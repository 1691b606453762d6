"""The checker of the tile lists (tests/util.py::assert_tile_lists) on hand-made lists: it must accept gsplat's lists and every
ordered sublist of them in the tight mode, and reject a foreign pair, a reordered tile and offsets that do not match the lists."""
import numpy as np
import pytest
import torch

from tests.util import assert_tile_lists


def _ref():
    # three tiles: [5, 2, 9] | [] | [7, 5]
    return {"isect_offsets": np.array([[[0, 3, 3]]], dtype=np.int32), "flatten_ids": np.array([5, 2, 9, 7, 5], dtype=np.int32),
            "isect_ids": np.array([10, 11, 12, 30, 31], dtype=np.int64)}


def _info(offsets, flat, ids, pad=0):
    n = len(flat)
    return {"isect_offsets": torch.tensor(offsets, dtype=torch.int32).view(1, 1, -1),
            "flatten_ids": torch.tensor(list(flat) + [123] * pad, dtype=torch.int32),
            "isect_ids": torch.tensor(list(ids) + [-1] * pad, dtype=torch.int64), "n_listed": torch.tensor(n, dtype=torch.int32)}


def test_gsplats_lists_and_ordered_sublists_pass():
    import mtgs_amd
    assert not mtgs_amd.lists_are_tight()      # gsplat's lists are the default
    with mtgs_amd.tight_lists():
        assert mtgs_amd.lists_are_tight()
        assert_tile_lists(_info([0, 3, 3], [5, 2, 9, 7, 5], [10, 11, 12, 30, 31]), _ref())
        assert_tile_lists(_info([0, 2, 2], [5, 9, 5], [10, 12, 31], pad=2), _ref())        # buffers keep gsplat's length
        assert_tile_lists(_info([0, 0, 0], [], [], pad=5), _ref())
    assert not mtgs_amd.lists_are_tight()


def test_list_mode_is_per_thread():
    """A viewer thread entering tight_lists() / exact_lists() must not flip the trainer thread's mode mid-frame
    (render_state_machine.py:142: two threads into the same path)."""
    import threading
    import mtgs_amd
    seen, go, done = {}, threading.Event(), threading.Event()

    def viewer():
        with mtgs_amd.tight_lists():
            seen["viewer_inside"] = mtgs_amd.lists_are_tight()
            go.set()
            done.wait(5.0)
        seen["viewer_after"] = mtgs_amd.lists_are_tight()

    t = threading.Thread(target=viewer)
    t.start()
    assert go.wait(5.0)
    seen["trainer_while_viewer_tight"] = mtgs_amd.lists_are_tight()
    with mtgs_amd.tight_lists():
        with mtgs_amd.exact_lists():
            seen["nested_exact"] = mtgs_amd.lists_are_tight()
        seen["back_to_tight"] = mtgs_amd.lists_are_tight()
    done.set()
    t.join()
    assert seen == {"viewer_inside": True, "trainer_while_viewer_tight": False, "nested_exact": False, "back_to_tight": True,
                    "viewer_after": False}


@pytest.fixture
def tight_mode():
    import mtgs_amd
    with mtgs_amd.tight_lists():
        yield


@pytest.mark.parametrize("offsets,flat,ids", [
    ([0, 2, 2], [9, 5, 5], [12, 10, 31]),        # a tile's pairs out of gsplat's order
    ([0, 2, 2], [5, 4, 5], [10, 11, 31]),        # a Gaussian gsplat does not list in that tile
    ([0, 1, 1], [5, 9, 5], [10, 12, 31]),        # offsets that put Gaussian 9 into the (empty) middle tile
    ([0, 2, 2], [5, 9, 5], [10, 12, 30]),        # the wrong isect_id (depth bits of another pair)
])
def test_wrong_lists_fail(offsets, flat, ids, tight_mode):
    with pytest.raises(AssertionError):
        assert_tile_lists(_info(offsets, flat, ids), _ref())


def test_exact_mode_demands_equality():
    import mtgs_amd
    for ctx in (mtgs_amd.exact_lists(), mtgs_amd.tight_lists(False)):      # (and the plain default)
        with ctx:
            assert_tile_lists(_info([0, 3, 3], [5, 2, 9, 7, 5], [10, 11, 12, 30, 31]), _ref())
            with pytest.raises(AssertionError):
                assert_tile_lists(_info([0, 2, 2], [5, 9, 5], [10, 12, 31]), _ref())
    assert_tile_lists(_info([0, 3, 3], [5, 2, 9, 7, 5], [10, 11, 12, 30, 31]), _ref())
    with pytest.raises(AssertionError):
        assert_tile_lists(_info([0, 2, 2], [5, 9, 5], [10, 12, 31]), _ref())
