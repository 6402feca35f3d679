"""The C-ABI shared library loads without a GPU and exports every function include/pt_amd.h
declares (no compute is called here)."""
import ctypes
import os
import re

import path_tracer_amd as pta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 35
    L = ctypes.CDLL(pta.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(pta.EXPORTS) == names, "binding list and header disagree"


def test_header_cites_the_reference_for_every_entry():
    text = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    assert text.count("src/cs488.h:") >= 25 and "src/main.cpp" in text


def test_null_and_host_only_contexts():
    L = pta.lib()
    assert L.pt_create(0, None) == pta.ERR_INVALID_ARG
    h = ctypes.c_void_p()
    assert L.pt_create(pta.DEVICE_NONE, ctypes.byref(h)) == pta.PT_OK
    assert L.pt_render(h, 1) == pta.ERR_NOT_READY
    assert L.pt_build(h, 0) == pta.ERR_NOT_READY  # no mesh yet
    assert b"no mesh" in L.pt_last_error(h)
    assert L.pt_set_option(h, 999, ctypes.c_uint64(0)) == pta.ERR_INVALID_ARG
    v = ctypes.c_uint64()
    assert L.pt_get_option(h, pta.OPT_MAX_LEVEL, ctypes.byref(v)) == pta.PT_OK and v.value == 15
    assert L.pt_get_option(h, pta.OPT_SEED, ctypes.byref(v)) == pta.PT_OK and v.value == 0xCAFEF00DD15EA5E5
    L.pt_destroy(h)


def test_struct_layout_matches_header():
    assert ctypes.sizeof(pta.PtMaterial) == 96 + 4 + 4 + 48 + 12 + 12


def test_python_constants_equal_the_header_enums():
    """Every PT_OPT_ / PT_CNT_ / PT_TIMER_ value of the header under the binding's name without the PT_ prefix."""
    text = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    pairs = re.findall(r"\bPT_((?:OPT|CNT|TIMER)_[A-Z_]+)\s*=\s*(\d+)", text)
    assert len(pairs) >= 30
    for name, value in pairs:
        assert getattr(pta, name) == int(value), name
    v = ctypes.c_uint64()
    h = ctypes.c_void_p()
    L = pta.lib()
    assert L.pt_create(pta.DEVICE_NONE, ctypes.byref(h)) == pta.PT_OK
    assert L.pt_get_option(h, pta.OPT_STREAMS, ctypes.byref(v)) == pta.PT_OK and v.value == 3
    assert L.pt_set_option(h, pta.OPT_STREAMS, ctypes.c_uint64(5)) == pta.ERR_INVALID_ARG
    L.pt_destroy(h)


def test_refill_threshold_above_a_wave_is_refused():
    """Bits 8..14 of PT_OPT_TRAVERSAL are the idle lanes a wave waits for before it fetches rays; a
    wave has 64, so a larger value could never be met and the walk would not end."""
    L = pta.lib()
    h = ctypes.c_void_p()
    v = ctypes.c_uint64()
    assert L.pt_create(pta.DEVICE_NONE, ctypes.byref(h)) == pta.PT_OK
    assert L.pt_set_option(h, pta.OPT_TRAVERSAL, ctypes.c_uint64(0x4041)) == pta.PT_OK   # 64: a whole wave idle
    assert L.pt_set_option(h, pta.OPT_TRAVERSAL, ctypes.c_uint64(0x4141)) == pta.ERR_INVALID_ARG  # 65
    assert L.pt_set_option(h, pta.OPT_TRAVERSAL, ctypes.c_uint64(0x7F01)) == pta.ERR_INVALID_ARG  # 127
    assert L.pt_set_option(h, pta.OPT_TRAVERSAL, ctypes.c_uint64(0x10001)) == pta.ERR_INVALID_ARG  # unknown bit
    assert L.pt_get_option(h, pta.OPT_TRAVERSAL, ctypes.byref(v)) == pta.PT_OK and v.value == 0x4041
    L.pt_destroy(h)
