"""The entry point that allocates nothing: fpn_shared_mlp_static_f32 (partial maxima of
a layer that pools more than 128 rows go to a buffer inside the library's device image)
must give fpn_shared_mlp_f32's results bit for bit, call after call, and refuse what its
buffer cannot hold.  Called through the C ABI with raw pointers."""
import ctypes

import numpy as np
import pytest
import torch

from fpn import _lib

pytestmark = pytest.mark.gpu


def _layer_inputs(dev, rows, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((rows, cin), generator=g).to(dev)
    w = (torch.randn((cin, cout), generator=g) * cin ** -0.5).to(dev)
    scale = (torch.rand((cout,), generator=g) + 0.5).to(dev)
    shift = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    return x, w, scale, shift


def _call_pair(lib, dev, rows, cin, cout, pool_rows, seed, relu=1):
    """(workspace entry point's result, static entry point's result)."""
    x, w, scale, shift = _layer_inputs(dev, rows, cin, cout, seed)
    groups = rows // pool_rows
    st = _lib.stream_of(x)
    need = lib.fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows)
    ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=dev)
    want = torch.full((groups, cout), float("nan"), device=dev)
    got = torch.full((groups, cout), float("nan"), device=dev)
    _lib.check(lib.fpn_shared_mlp_f32(rows, cin, cout, _lib.ptr(x), cin, _lib.ptr(w), _lib.ptr(scale),
                                      _lib.ptr(shift), None, 0, relu, pool_rows, _lib.ptr(want), cout,
                                      _lib.ptr(ws), need, st))
    _lib.check(lib.fpn_shared_mlp_static_f32(rows, cin, cout, _lib.ptr(x), cin, _lib.ptr(w), _lib.ptr(scale),
                                             _lib.ptr(shift), None, 0, relu, pool_rows, _lib.ptr(got), cout, st))
    torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy()


def test_static_scratch_entry_point_matches_workspace_entry_point(device):
    lib = _lib.load()
    # first call: 48 frustums x 1024 points, 64 -> 256, max over each frustum's points
    want, got = _call_pair(lib, device, 48 * 1024, 64, 256, 1024, seed=1)
    assert np.isfinite(want).all()
    np.testing.assert_array_equal(got, want)
    # second call, smaller, straight after: the buffer still holds the first call's maxima
    want, got = _call_pair(lib, device, 5 * 512, 35, 96, 512, seed=2, relu=0)
    assert np.isfinite(want).all()
    np.testing.assert_array_equal(got, want)
    # a layer that needs no scratch at all goes through the same entry point
    want, got = _call_pair(lib, device, 12 * 64, 16, 40, 64, seed=3)
    np.testing.assert_array_equal(got, want)


def _static_call(lib, x, w, scale, shift, rows, cin, cout, pool_rows, out, relu=1):
    _lib.check(lib.fpn_shared_mlp_static_f32(rows, cin, cout, _lib.ptr(x), cin, _lib.ptr(w), _lib.ptr(scale),
                                             _lib.ptr(shift), None, 0, relu, pool_rows, _lib.ptr(out), cout,
                                             _lib.stream_of(x)))


def test_static_scratch_entry_point_allocates_nothing(device):
    """Its defining property: a (warmed) call changes neither the device's free memory nor
    the number of device allocations the process has made."""
    lib = _lib.load()
    rows, cin, cout, pool = 24 * 1024, 64, 256, 1024
    x, w, scale, shift = _layer_inputs(device, rows, cin, cout, seed=5)
    out = torch.empty((rows // pool, cout), device=device)
    _static_call(lib, x, w, scale, shift, rows, cin, cout, pool, out)       # first call on the device: address look-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(device)
    allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    segs0 = torch.cuda.memory_stats(device).get("segment.all.allocated", 0)
    for _ in range(3):
        _static_call(lib, x, w, scale, shift, rows, cin, cout, pool, out)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(device)
    assert free1 == free0, "device memory changed by %d bytes" % (free0 - free1)
    assert torch.cuda.memory_stats(device).get("num_device_alloc", 0) == allocs0
    assert torch.cuda.memory_stats(device).get("segment.all.allocated", 0) == segs0


def test_static_scratch_calls_back_to_back_on_one_stream(device):
    """Two calls of different shapes queued on one stream with no synchronisation between
    them share the buffer in stream order: the second starts clearing / using it only after
    the first's kernels have finished with it.  Both equal the workspace entry point."""
    lib = _lib.load()
    shapes = ((40 * 1024, 64, 256, 1024, 6), (9 * 256, 48, 72, 256, 7))
    ins = [_layer_inputs(device, r, ci, co, seed) for r, ci, co, _, seed in shapes]
    outs = [torch.full((r // pool, co), float("nan"), device=device) for r, _, co, pool, _ in shapes]
    torch.cuda.synchronize()
    for (r, ci, co, pool, _), (x, w, scale, shift), out in zip(shapes, ins, outs):
        _static_call(lib, x, w, scale, shift, r, ci, co, pool, out)          # queued, not waited for
    torch.cuda.synchronize()
    for (r, ci, co, pool, _), (x, w, scale, shift), out in zip(shapes, ins, outs):
        need = lib.fpn_shared_mlp_workspace_bytes(r, co, pool)
        ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=device)
        want = torch.empty_like(out)
        _lib.check(lib.fpn_shared_mlp_f32(r, ci, co, _lib.ptr(x), ci, _lib.ptr(w), _lib.ptr(scale), _lib.ptr(shift), None, 0,
                                          1, pool, _lib.ptr(want), co, _lib.ptr(ws), need, _lib.stream_of(x)))
        torch.cuda.synchronize()
        assert torch.isfinite(want).all()
        assert torch.equal(out, want)


def test_static_scratch_entry_point_refuses_oversized_requests(device):
    lib = _lib.load()
    rows, cout, pool = 4096 * 1024, 1024, 1024          # 32768 row blocks x 1024 floats = 128 MiB
    assert lib.fpn_shared_mlp_workspace_bytes(rows, cout, pool) > _lib.STATIC_SCRATCH_BYTES
    p = ctypes.c_void_p(256)                            # never dereferenced: refused before any launch
    rc = lib.fpn_shared_mlp_static_f32(rows, 64, cout, p, 64, p, None, None, None, 0, 1, pool, p, cout, None)
    assert rc == -1
    assert b"static buffer holds" in lib.fpn_last_error_string()
    # the largest request that fits is accepted (8 MiB = 2048 row blocks x 1024 columns)
    rows = 2048 * 128
    assert lib.fpn_shared_mlp_workspace_bytes(rows, 1024, 1024) == _lib.STATIC_SCRATCH_BYTES
    want, got = _call_pair(lib, device, rows, 8, 1024, 1024, seed=4)
    np.testing.assert_array_equal(got, want)
