"""CPU suite: the drop-in boundary stays consistent — include/gaz_engine.h, the ctypes mirror in grok_alpha_zero_amd/engine.py,
the stub INTEGRATION.md shows a maintainer, and what the built libraries export.  (the doc's struct had drifted three
fields behind the header and nothing guarded the struct size.)"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_stub  # noqa: E402
from grok_alpha_zero_amd import engine as E  # noqa: E402

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "libgaz_emu.so")
HDR = os.path.join(ROOT, "include", "gaz_engine.h")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU


def _ctypes_fields(cls):
    out = []
    for name, t in cls._fields_:
        n = 0
        if hasattr(t, "_length_"):
            n, t = t._length_, t._type_
        out.append((name, t, n))
    return out


@pytest.mark.parametrize("c_name,cls", [("gaz_engine_config", E.EngineConfig), ("gaz_search_hyperparams", E.SearchHyperparams),
                                        ("gaz_tensor", E.Tensor), ("gaz_record_layout", E.RecordLayout)])
def test_ctypes_mirror_matches_header(c_name, cls):
    hdr = abi_stub.parse_structs()[c_name]
    want = []
    for f, t, n in hdr:
        ct = getattr(C, abi_stub.CTYPES[t]) if t in abi_stub.CTYPES else {"char*": C.c_char_p, "float*": C.POINTER(C.c_float)}[t]
        want.append((f, ct, n))
    assert _ctypes_fields(cls) == want


def test_struct_sizes_match_a_c_compiler(tmp_path):
    """sizeof / offsetof from gcc on the header itself == the ctypes layout."""
    fields = [f for f, _ in E.EngineConfig._fields_]
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gaz_engine.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(gaz_engine_config),'
                   ' sizeof(gaz_search_hyperparams), sizeof(gaz_tensor), sizeof(gaz_record_layout));'
                   + "".join(f'printf("%zu\\n", offsetof(gaz_engine_config, {f}));' for f in fields) + "return 0;}\n")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(x) for x in out[:4]] == [C.sizeof(E.EngineConfig), C.sizeof(E.SearchHyperparams), C.sizeof(E.Tensor), C.sizeof(E.RecordLayout)]
    assert [int(x) for x in out[4:]] == [getattr(E.EngineConfig, f).offset for f in fields]


def test_integration_md_stub_is_the_generated_one():
    s = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a, b = s.index(abi_stub.BEGIN), s.index(abi_stub.END)
    block = s[a + len(abi_stub.BEGIN):b].strip()
    assert block == "```python\n" + abi_stub.stub_text() + "\n```", "run: python tools/abi_stub.py --write"
    # and the snippet around it really is a complete, loadable binding of the struct: exec the classes and compare layouts
    ns = {"C": C}
    exec(abi_stub.stub_text(), ns)
    assert C.sizeof(ns["Cfg"]) == C.sizeof(E.EngineConfig) and C.sizeof(ns["Hyper"]) == C.sizeof(E.SearchHyperparams)
    assert [f for f, _ in ns["Cfg"]._fields_] == [f for f, _ in E.EngineConfig._fields_]
    assert "struct_size=C.sizeof(Cfg)" in s and f"gaz_engine_abi_version() == {E.ABI_VERSION}" in s


def test_abi_version_constant_matches_header():
    m = re.search(r"#define\s+GAZ_ENGINE_ABI_VERSION\s+(\d+)", open(HDR).read())
    assert int(m.group(1)) == E.ABI_VERSION


def _exports(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_libraries_export_every_declared_symbol(emu_lib):
    """Every function include/gaz_engine.h declares is exported by the emulation build and — when it has been built (build() does
    that in the driver's build check) — by the hipcc product library.  No compute calls here."""
    want = set(abi_stub.exported_functions())
    assert len(want) >= 30
    assert want <= _exports(emu_lib), sorted(want - _exports(emu_lib))
    if os.path.exists(E.DEFAULT_LIB):
        got = _exports(E.DEFAULT_LIB)
        assert want <= got, sorted(want - got)
        L = E.load_library()                      # resolves every symbol + checks ABI version / config size
        assert L.gaz_engine_abi_version() == E.ABI_VERSION


def test_create_rejects_a_stale_struct(emu_lib):
    """A binding built against an older header (shorter struct, or no struct_size at all) is refused with a last_error text."""
    L = E.load_library(emu_lib)
    cfg = E.EngineConfig(struct_size=C.sizeof(E.EngineConfig) - 12, game=1, n_games=2, run_iterations=8, max_actions=42)
    h = C.c_void_p()
    assert L.gaz_engine_create(C.byref(cfg), C.byref(h)) != 0 and not h.value
    msg = L.gaz_engine_last_error(None).decode()
    assert "struct_size" in msg and str(C.sizeof(E.EngineConfig)) in msg
    cfg.struct_size = 1                           # the r1 layout started with `game`: GAZ_GAME_CONNECT4 lands in struct_size
    assert L.gaz_engine_create(C.byref(cfg), C.byref(h)) != 0
    cfg.struct_size = C.sizeof(E.EngineConfig); cfg.tau = -1.0; cfg.c_puct_base = 19652.0; cfg.c_puct_init = 2.5
    assert L.gaz_engine_create(C.byref(cfg), C.byref(h)) == 0, L.gaz_engine_last_error(None)
    hp = E.SearchHyperparams(struct_size=8)
    assert L.gaz_engine_set_hyperparams(h, C.byref(hp)) != 0 and b"struct_size" in L.gaz_engine_last_error(h)
    L.gaz_engine_destroy(h)


def test_probe_rules_rejects_bad_input(emu_lib):
    eng = E.SelfPlayEngine("Connect4", 2, 8, 42, 0, 0, 2.5, 0.5, seed=1, lib_path=emu_lib)
    with pytest.raises(E.EngineError):
        eng.probe_rules([[9]])                    # column out of range: rejected on the host, never reaches a kernel
    r = eng.probe_rules([[0] * 7])               # seventh stone in a six-row column: flagged by the device rule code
    assert r["winner"][0] == -99
    eng.close()
