"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from hanabizero_amd._lib import LIB_PATH, declared_symbols
    lib = ctypes.CDLL(LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_error_path_without_gpu_calls():
    from hanabizero_amd._lib import lib
    assert lib.hz_version() >= 1
    h = ctypes.c_void_p()
    assert lib.hz_tree_create(ctypes.byref(h), 0, 20, 50, 0) == -1  # argument validation happens before any HIP call
    assert b"num_trees" in lib.hz_last_error()
    assert lib.hz_tree_create(ctypes.byref(h), 4, 65, 50, 0) == -1
    assert b"num_actions" in lib.hz_last_error()


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hanabizero_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                for needle in ("libhz_oracle", "libref_tree", "libpyhanabi", "oracle.cport", "oracle.ref", '#include "../oracle',
                               "oracle/_ref"):
                    assert needle not in text, (f, needle)
