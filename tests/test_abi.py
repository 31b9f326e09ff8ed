"""The C-ABI library loads without a GPU and exports every symbol include/kiss_hip.h declares."""
import ctypes
import os
import re

import kiss_amd
from kiss_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kiss_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kiss_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = kiss_amd.load()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(_lib.EXPORTED_SYMBOLS) == syms


def test_version_and_strerror():
    lib = kiss_amd.load()
    assert lib.kiss_hip_version() == 103
    assert _lib.strerror(0) == "ok"
    assert "invalid" in _lib.strerror(-1)


def test_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        return
    lib = kiss_amd.load()
    ctx = ctypes.c_void_p()
    rc = lib.kiss_hip_ctx_create(ctypes.byref(ctx), 0, 1000)
    assert rc == -2  # KISS_HIP_E_NO_DEVICE: no silent CPU fallback
    try:
        kiss_amd.KISS1Sorter.get_suffix_array_dna([0, 1, 2, 3], 256)
        raised = False
    except kiss_amd.KissHipError:
        raised = True
    assert raised


def _sizeof_from_header(struct_name):
    """compiles a one-line C program against include/kiss_hip.h and returns sizeof(struct) as the C compiler sees it"""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "kiss_hip.h"\nint main(void){printf("%%zu", sizeof(%s));return 0;}\n'
                             % struct_name)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        return int(subprocess.check_output([exe]).decode())


def test_ctypes_structs_match_the_header():
    # the ctypes mirrors must have exactly the size the C compiler gives the header's structs
    assert ctypes.sizeof(_lib.Stats) == _sizeof_from_header("kiss_hip_stats")
    assert ctypes.sizeof(_lib.VerifyReport) == _sizeof_from_header("kiss_hip_verify_report")
    assert ctypes.sizeof(_lib.FmiView) == _sizeof_from_header("kiss_hip_fmi_view")
    assert ctypes.sizeof(_lib.MultiStats) == _sizeof_from_header("kiss_hip_multi_stats")
    # and the field the Python side reads last sits where the header puts it
    assert _lib.Stats.ms_refine.offset == ctypes.sizeof(_lib.Stats) - 32 and _lib.Stats.tie_run_retries.offset == ctypes.sizeof(_lib.Stats) - 8


def test_cpp_host_facade_compiles_and_links(tmp_path):
    # the header-only C++ facade satisfies the reference's SASorter shape and links against the library
    import subprocess
    exe = tmp_path / "hfc"
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "host_facade_check.cpp"), "-o",
           str(exe), "-L" + os.path.join(ROOT, "kiss_amd"), "-lkiss_hip", "-Wl,-rpath," + os.path.join(ROOT, "kiss_amd")]
    subprocess.check_call(cmd)
    assert subprocess.call([str(exe)]) == 0


def test_shipped_library_has_no_environment_hooks():
    # the default libkiss_hip.so looks at the environment once per context and only for three host-side
    # knobs; every A-B switch, tuning value, fault injection and trace exists in libkiss_hip_hooks.so (-DKISS_HIP_HOOKS) only
    import subprocess
    default, hooks = kiss_amd.load(False), kiss_amd.load(True)
    assert default.kiss_hip_has_hooks() == 0 and hooks.kiss_hip_has_hooks() == 1
    for s in _lib.EXPORTED_SYMBOLS:
        assert hasattr(hooks, s), "hooks build misses export " + s
    names = set(re.findall(rb"KISS_HIP_[A-Z0-9_]+", open(_lib.LIB_PATH, "rb").read()))
    assert names == {b"KISS_HIP_DEBUG", b"KISS_HIP_XFER_THREADS", b"KISS_HIP_PREFAULT_THREADS"}, sorted(names)
    assert b"KISS_HIP_NO_SERIALIZE" in open(_lib.HOOKS_LIB_PATH, "rb").read()
    # and in the sources: getenv only in api.hip's two option readers (one of them compiled into the hooks build only)
    csrc = os.path.join(ROOT, "kiss_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".hpp")):
            continue
        text = open(os.path.join(csrc, fn)).read()
        if fn != "api.hip":
            assert "getenv" not in text, fn
            continue
        body = re.sub(r"#ifdef KISS_HIP_HOOKS.*?#endif", "", text, flags=re.S)
        assert body.count("getenv(") == 3, body.count("getenv(")  # KISS_HIP_DEBUG / _XFER_THREADS / _PREFAULT_THREADS
