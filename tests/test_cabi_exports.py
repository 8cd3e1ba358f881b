"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/fsraft.h declares, the ctypes table covers them all, and the product path fails loudly
(no CPU / eager fallback) when it cannot run on the HIP library."""
import argparse
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(which=("fsraft.h", "fsraft_tuning.h")):
    """fsraft.h = the drop-in boundary; fsraft_tuning.h = measurement / test knobs (not part of the boundary)."""
    txt = "".join(open(os.path.join(ROOT, "include", h)).read() for h in which)
    return sorted(set(re.findall(r"^int (fsraft_\w+)\(", txt, flags=re.M)))


def test_boundary_header_carries_no_tuning_knobs():
    names = header_symbols(("fsraft.h",))
    assert not [n for n in names if "tuning" in n or n.startswith("fsraft_set_") and n != "fsraft_set_arithmetic"], names
    from flow_supervisor_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    if os.path.basename(_lib.LIB_PATH) == "libfsraft.so":
        lib.fsraft_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
        for key in (24, 25, 30):               # experiment kernels are not in the shipped library
            assert lib.fsraft_set_tuning(key, 0) == 1


def test_library_exports_every_declared_symbol():
    from flow_supervisor_amd import _lib
    names = header_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    _lib.load()


def test_conv_desc_layout_matches_header():
    """Field order of the ctypes mirror == field order of struct fsraft_conv_desc."""
    from flow_supervisor_amd import _lib
    txt = open(os.path.join(ROOT, "include", "fsraft.h")).read()
    body = txt[txt.index("typedef struct fsraft_conv_desc {"):txt.index("} fsraft_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.sub(r"\[\d+\]", "", part.split()[-1].lstrip("*")))
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]


def test_ktot_host_logic():
    """Packed-K size rule used on both sides of the ABI (no GPU work: pure host arithmetic)."""
    from flow_supervisor_amd import ops
    assert ops.conv_ktot([324], 1, 1) == 352
    assert ops.conv_ktot([128, 128, 128], 1, 5) == 1920
    assert ops.conv_ktot([96, 64, 82], 3, 3) == 9 * (96 + 64 + 96)
    assert ops.pyramid_sizes(55, 128) == [(55, 128), (27, 64), (13, 32), (6, 16)]


def test_hot_path_refuses_cpu_tensors():
    from flow_supervisor_amd.core.corr import CorrBlock
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.core.update import BasicUpdateBlock
    a = argparse.Namespace(small=False, corr_levels=4, corr_radius=4)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        CorrBlock(torch.zeros(1, 8, 16, 16), torch.zeros(1, 8, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        BasicUpdateBlock(a)(torch.zeros(1, 128, 8, 8), torch.zeros(1, 128, 8, 8), torch.zeros(1, 324, 8, 8), torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError):
        RAFT(argparse.Namespace(small=False))(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))


def test_missing_library_is_loud(monkeypatch):
    from flow_supervisor_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfsraft.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.load()


def test_state_dict_keys_match_reference():
    import json
    from flow_supervisor_amd.core.raft import RAFT
    for small in (False, True):
        m = RAFT(argparse.Namespace(small=small))
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", f"raft_{'small' if small else 'basic'}_shapes.json")))
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref


def test_l2l_and_gma_state_dict_keys_match_reference():
    import json
    from flow_supervisor_amd.core.gma_l2l import GMAL2L
    from flow_supervisor_amd.core.gma_network import RAFTGMA
    from flow_supervisor_amd.core.l2l import L2L
    gd = os.path.join(ROOT, "tests", "golden")
    m = L2L(argparse.Namespace(small=False))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == json.load(open(os.path.join(gd, "l2l_basic_shapes.json")))
    a = argparse.Namespace(num_heads=1, position_only=False, position_and_content=False)
    ref = json.load(open(os.path.join(gd, "raft_gma_shapes.json")))
    got = {k: list(v.shape) for k, v in RAFTGMA(a).state_dict().items() if not k.endswith("rel_ind")}
    assert got == ref
    l2l = {k for k in GMAL2L(a).state_dict() if not k.endswith("rel_ind")}
    assert l2l == set(ref) | {"grad_" + k for k in ref if k.startswith("update_block.")}


def test_product_never_imports_oracle():
    """The oracle is test infrastructure; nothing shipped may import it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "flow_supervisor_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(d, f)).read(), flags=re.M):
                bad.append(f)
    assert not bad, bad


def test_entry_points_reject_bad_arguments_before_touching_the_gpu():
    """Argument validation is host code: null pointers, sizes beyond the kernels' bitmaps and unknown switch values come back
    as FS_ERR_ARG (1) without a HIP call -- runs on the CPU-only box.  (Round-3 entry points; the older ones behave alike.)"""
    import ctypes
    from flow_supervisor_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(None)
    nullpp = ctypes.cast(null, _lib._PP)
    strides = (ctypes.c_int64 * 3)(0, 0, 0)
    assert lib.fsraft_corr_bwd_ktiles(nullpp, strides, 1, 4, 1, 16, 16, 4, 1, 0, 0, null, null, 0, null, null, null, 0, null, None) == 1
    assert lib.fsraft_gemm_rec_nt_list(null, 0, 0, null, 0, 0, null, 0, 0, 1, 32, 32, 32, 1.0, 1, 0, null, null, 0, 0, null, null, None) == 1
    assert lib.fsraft_gemm_rec_tn_list(null, 0, 0, null, 0, 0, null, 0, 0, 1, 32, 32, 32, 1.0, 1, 0, null, null, 0, 0, null, null, None) == 1
    assert lib.fsraft_gru_bwd1(null, null, null, null, null, null, 128, null, null, null, null, 16, 128, null, null, null, None) == 1
    assert lib.fsraft_set_lookup_policy(3) == 1 and lib.fsraft_set_lookup_policy(-1) == 0
    assert lib.fsraft_set_ktile_exact(5) == 1 and lib.fsraft_set_ktile_exact(0) == 0
    assert lib.fsraft_conv_workspace(ctypes.c_void_p(8), 16) == 1          # misaligned scratch
    assert lib.fsraft_conv_workspace(null, 0) == 0
    assert lib.fsraft_abi_version() == 6
    assert lib.fsraft_amax_scaled(null, 1.0, null, None) == 1 and lib.fsraft_amax(null, 1, 1, 1, null, None) == 1
