"""The C-ABI library loads and exports every symbol include/dyk_hip.h declares, the ctypes binding
covers exactly those symbols, and the ctypes structure layouts agree with the C compiler's
(sizeof / offsetof checked through a tiny gcc-built probe).  No compute, no GPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dyk_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dyk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dyk import lib
    h = lib.load()
    names = _declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(h, n), "libdyk_hip.so does not export %s" % n
    assert sorted(lib.SIGNATURES) == names, "ctypes table and header disagree: %s" % (
        set(lib.SIGNATURES) ^ set(names))
    assert h.dyk_abi_version() == 5
    assert h.dyk_error_string(0) == b"ok" and h.dyk_error_string(-1) != b"ok"


def test_null_descriptors_are_rejected_without_a_gpu():
    from dyk import lib
    h = lib.load()
    assert h.dyk_conv_igemm(None, None) == -1
    assert h.dyk_conv_wgrad(None, None) == -1
    assert h.dyk_run_commands(None, 0, None, None) == -1
    d = lib.DykConvDesc()
    assert h.dyk_conv_igemm(ctypes.byref(d), None) == -1          # null pointers inside


STRUCTS = ["DykConvDesc", "DykWgradDesc", "DykEwDesc", "DykBnFinalizeDesc", "DykSeFcDesc", "DykTransposeEntry", "DykPadEntry",
           "DykMiscDesc", "DykCommand", "DykDwDesc", "DykGradReduceEntry", "DykDecodeDesc", "DykTargetsDesc", "DykLossDesc", "DykNmsDesc", "DykOptimDesc", "DykSchedEntry", "DykStemDesc"]


def test_struct_layouts_match_the_c_compiler(tmp_path):
    from dyk import lib
    src = tmp_path / "probe.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dyk_hip.h"', "int main(void){"]
    fields = {}
    for s in STRUCTS:
        cls = getattr(lib, s)
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        last = cls._fields_[-1][0]
        fields[s] = last
        lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, last, s, last))
    lines.append("return 0;}")
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for s in STRUCTS:
        cls = getattr(lib, s)
        assert int(out[s]) == ctypes.sizeof(cls), "sizeof(%s): C %s vs ctypes %d" % (s, out[s], ctypes.sizeof(cls))
        assert int(out["%s.%s" % (s, fields[s])]) == getattr(cls, fields[s]).offset, s


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from dyk import lib
    monkeypatch.setattr(lib, "_lib", None)
    with pytest.raises(lib.DykLibraryError):
        lib.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_refused():
    import torch
    from build_utils.parse_config import materialize_cfg
    from dyk import lib, ops
    from models import YOLO
    with pytest.raises(lib.DykError):
        ops.to_nhwc(torch.zeros(1, 8, 4, 4), torch.float32)
    m = YOLO(materialize_cfg("kaist_yolov3"))
    with pytest.raises(lib.DykError):
        m(torch.zeros(1, 3, 64, 64))


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """the digest of the kernel sources is compiled into the library (Makefile -> build_sha.h -> dyk_build_sha()); the loader
    recomputes it from the tree: a stale .so with the right ABI version does not load"""
    from dyk import buildinfo, lib
    h = lib.load()
    assert h.dyk_build_sha().decode() == buildinfo.native_sha() and len(buildinfo.native_sha()) == 16
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(buildinfo, "native_sha", lambda: "0" * 16)
    monkeypatch.delenv("DYK_LIB", raising=False)
    with pytest.raises(lib.DykLibraryError, match="built from other sources"):
        lib.load()
    monkeypatch.setenv("DYK_ALLOW_STALE_LIB", "1")          # the documented override
    assert lib.load() is not None
