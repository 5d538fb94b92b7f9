"""CPU-only: the C-ABI library is built, loads, and exports every symbol include/melo_gan_hip.h
declares (no compute calls without a GPU); the ctypes signature table covers the header."""
import os
import re

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_functions():
    src = open(os.path.join(ROOT, "include", "melo_gan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.mg_version() >= 100
    assert set(_lib.SIGNATURES) <= set(names), set(_lib.SIGNATURES) - set(names)


def test_bad_arguments_are_rejected_on_the_host():
    """Argument validation happens before any launch, so it can be exercised without a GPU."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import _lib
    lib = _lib.load()
    rc = lib.mg_conv1d_gather(None, None, None, 1, 1, 1, 1, 5, 1, 0, 5, 1, 0, 0, None, None, 0, None)
    assert rc == -1 and b"null" in lib.mg_last_error()
    rc = lib.mg_wgrad(None, None, 0, None, None, 0, None, None, 0, 1, 1, 1, 1, 1, 1, None, 0, None)
    assert rc == -1
    assert lib.mg_wgrad_workspace_bytes(64, 64, 5, 8, 32) > 0
    # the multi-job entry points: job count limits, null tensors, malformed rows
    assert lib.mg_wgrad_multi(None, 1, 1, 1, None, 0, None) == -1
    jobs = (_lib.WgradJob * 1)()
    assert lib.mg_wgrad_multi(jobs, 0, 1, 1, None, 0, None) == -1
    assert lib.mg_wgrad_multi(jobs, _lib.MAX_WGRAD_JOBS + 1, 1, 1, None, 0, None) == -1
    assert lib.mg_wgrad_multi(jobs, 1, 1, 1, None, 0, None) == -1 and b"segment 0" in lib.mg_last_error()
    st = (_lib.StageJob * 1)()
    assert lib.mg_stage_rows(st, 0, 4, None) == -1 and lib.mg_stage_rows(st, _lib.MAX_STAGE_JOBS + 1, 4, None) == -1
    assert lib.mg_stage_rows(st, 1, 4, None) == -1                      # null source / destination
    st[0].src, st[0].dst, st[0].row_bytes, st[0].src_rows = 256, 512, 6, 8
    assert lib.mg_stage_rows(st, 1, 4, None) == -1                      # rows must be multiples of 4 bytes
    st[0].row_bytes, st[0].dst_pitch = 8, 4
    assert lib.mg_stage_rows(st, 1, 4, None) == -1 and b"dst_pitch" in lib.mg_last_error()
    st[0].dst_pitch, st[0].src_rows = 0, 2
    assert lib.mg_stage_rows(st, 1, 4, None) == -1                      # unindexed source shorter than the batch
    assert lib.mg_dhead_fwd_bwd(None, None, None, None, None, None, None, None, 4, 4, 8, 8, 0, None) == -1


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    with pytest.raises(ValueError):
        ops.conv1d_fwd(torch.zeros(1, 8, 4), torch.zeros(8, 4, 5), torch.zeros(1, 4, 8), 2)
