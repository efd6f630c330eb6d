"""The C-ABI library loads on a machine without a GPU and exports every symbol include/tspgnn.h
declares; argument validation answers before any launch."""
import ctypes
import os
import re

from conftest import ROOT
from tspgnn import _lib

HEADER = os.path.join(ROOT, "include", "tspgnn.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|long long|double|const char\*)\s+(tspgnn_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = args
    return out


def test_every_declared_symbol_is_exported_and_bound():
    decl = declared_functions()
    assert len(decl) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, args in decl.items():
        assert hasattr(lib, name), "libtspgnn.so lacks %s" % name
        if name in ("tspgnn_version", "tspgnn_last_error") or name in _lib.HOST_FUNCTIONS:
            continue
        table = _lib.SIZE_QUERIES if name in _lib.SIZE_QUERIES else _lib.SIGNATURES
        assert name in table, "no ctypes signature for %s" % name
        assert len(table[name]) == len(args), name
    for name in list(_lib.SIGNATURES) + list(_lib.SIZE_QUERIES):
        assert name in decl, "%s bound but not declared in tspgnn.h" % name


def test_abi_version_and_struct_layouts_agree(tmp_path):
    """Header, library and ctypes binding carry the same ABI version, and every task structure has the same size and
    field offsets in C (gcc on include/tspgnn.h) and in the binding -- a layout change without a version bump would
    hand the kernels garbage pointers."""
    import subprocess
    hv = int(re.search(r"#define\s+TSPGNN_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert hv == _lib.ABI_VERSION == _lib.lib.tspgnn_version()
    pairs = {"tspgnn_mlp_task": _lib.MlpTask, "tspgnn_lstm_task": _lib.LstmTask, "tspgnn_cell_mlp_task": _lib.CellMlpTask,
             "tspgnn_mlp_task_bf16": _lib.MlpTaskB, "tspgnn_lstm_task_bf16": _lib.LstmTaskB,
             "tspgnn_lstm_bwd_task": _lib.LstmBwdTask, "tspgnn_mlp_bwd_task": _lib.MlpBwdTask, "tspgnn_mlp_bwd_rc_task": _lib.MlpBwdRcTask,
             "tspgnn_mp_loop_args": _lib.MpLoopArgs, "tspgnn_mp_resident_args": _lib.MpResidentArgs}
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "tspgnn.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        src.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            if fname == "cell":
                continue
            src.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        src.append('printf("\\n");')
    src += ["return 0; }"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(c), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().strip().splitlines()
    for line in lines:
        parts = line.split()
        cls = pairs[parts[0]]
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_ if f != "cell"]
        assert [int(x) for x in parts[1:]] == want, parts[0]


def test_version_and_error_string():
    assert _lib.lib.tspgnn_version() == _lib.ABI_VERSION == 5
    status = _lib.lib.tspgnn_gather2_sum_f32(None, None, None, 4, 4, 3, None)   # d=3: rejected before launch
    assert status == -1
    assert b"multiple of 4" in _lib.lib.tspgnn_last_error()
    try:
        _lib.call("tspgnn_mlp_fwd_f32", None, None, None, None, 0, 8, 48, 4, 7, None)
        assert False
    except _lib.TspgnnError as e:
        assert e.status == -1 and "d=48" in str(e)
    assert _lib.lib.tspgnn_lnlstm_fwd_f32(None, 8, None, None, None, None, None, None, 4, 64, None) == -1
    # empty problems are a no-op, not an error
    assert _lib.lib.tspgnn_gather2_sum_f32(None, None, None, 0, 0, 64, None) == 0
    assert _lib.lib.tspgnn_segment_mean_f32(None, None, None, 0, None) == 0
