"""The C-ABI library loads without a GPU and exports every symbol include/ddp_hip/ddp_hip.h declares."""
import ctypes
import os
import re

from ddp_pinocchio_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    L = capi.lib()
    header = open(os.path.join(ROOT, "include", "ddp_hip", "ddp_hip.h")).read()
    declared = set(re.findall(r"\b(ddp_hip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"libddp_hip.so does not export {name}"
    assert L.ddp_hip_abi_version() == 3


def test_builtin_models_are_deterministic():
    a = capi.BuiltinModel(capi.BUILTIN_TREE38, seed=7)
    b = capi.BuiltinModel(capi.BUILTIN_TREE38, seed=7)
    c = capi.BuiltinModel(capi.BUILTIN_TREE38, seed=8)
    assert a.nv == 38 and (a.parent < range(38)).all()
    assert (a.mass_j == b.mass_j).all() and (a.Ic == b.Ic).all()
    assert not (a.mass_j == c.mass_j).all()
    assert capi.BuiltinModel(capi.BUILTIN_CHAIN6).nv == 6
    assert capi.BuiltinModel(capi.BUILTIN_PENDULUM).nv == 1


def test_no_device_fails_loudly():
    # on a box without a GPU ddp_hip_create must refuse (no CPU fallback)
    if capi.lib().ddp_hip_device_count() > 0:
        return
    spec = capi.ProblemSpec(capi.BuiltinModel(capi.BUILTIN_PENDULUM), T=5)
    try:
        capi.Context(spec)
    except capi.DdpHipError as e:
        assert e.code == capi.E_NODEVICE
    else:
        raise AssertionError("ddp_hip_create succeeded without a device")


def test_bad_arguments_are_rejected():
    L = capi.lib()
    assert L.ddp_hip_create(None, 0, 0, None) == capi.E_ARG
    assert L.ddp_hip_seq_size(None, 0) == -1
    assert L.ddp_hip_strerror(capi.E_NODEVICE)
