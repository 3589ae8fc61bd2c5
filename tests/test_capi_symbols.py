"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/dynfu_amd.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (torch's bundled HIP runtime must be the one the library binds to)
    from dynfu_amd import build as B
    return ctypes.CDLL(B.build())


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dynfu_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfa_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from dynfu_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdynfu_amd.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names  # the Python binding covers the whole header


def test_version_and_error_string(lib):
    lib.dfa_version.restype = ctypes.c_char_p
    lib.dfa_last_error.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dfa_version()
    # argument validation happens before any HIP call, so it is testable without a GPU
    lib.dfa_tsdf_clear.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert lib.dfa_tsdf_clear(None, 8, 8, 8, None) == 1  # DFA_ERR_INVALID
    assert b"bad volume" in lib.dfa_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    import dynfu_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dynfu_amd.DynfuAmdError):
        dynfu_amd.Solver(8, 8, 4)
    with pytest.raises(dynfu_amd.DynfuAmdError):
        dynfu_amd.tsdf_clear(torch.zeros((4, 4, 4), dtype=torch.int32))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dynfu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text and "orc_" not in text, f


def test_product_library_has_no_ab_switches():
    """A/B scaffolding is not product: every DFA_* environment switch goes through dev_env() (csrc/dev_switch.hpp), which is
    getenv only under -DDFA_DEV_AB (libdynfu_amd_dev.so); the product library reads ONE variable, once per plan."""
    csrc = os.path.join(ROOT, "dynfu_amd", "csrc")
    sites = []
    for f in sorted(os.listdir(csrc)):
        if f == "dev_switch.hpp":
            continue
        for n, line in enumerate(open(os.path.join(csrc, f)), 1):
            code = line.split("//")[0]
            if re.search(r"(?<![_a-z])getenv\s*\(", code):
                sites.append((f, n, code.strip()))
    assert [s[2] for s in sites] == ['const char* e    = getenv("DFA_ASSEMBLE_DETERMINISTIC");'], sites
    # the product binary does not contain the names of the development switches
    blob = open(os.path.join(ROOT, "dynfu_amd", "libdynfu_amd.so"), "rb").read()
    for name in (b"DFA_PCG_VARIANT", b"DFA_TSDF_LEGACY", b"DFA_TSDF_ZCHUNK", b"DFA_MB_FORM", b"DFA_GRID_FOUR_KERNELS",
                 b"DFA_TSDF_VX4", b"DFA_NO_REGRADIENT", b"DFA_S6_NO_GRAPH"):
        assert name not in blob, name


def test_development_flavour_exports_the_same_abi():
    import torch  # noqa: F401
    from dynfu_amd import build as B
    dev = ctypes.CDLL(B.build(dev=True))
    for n in _declared_symbols():
        assert hasattr(dev, n), "libdynfu_amd_dev.so does not export %s" % n
    blob = open(B.DEV_LIB, "rb").read()
    assert b"DFA_PCG_VARIANT" in blob and b"DFA_TSDF_ZCHUNK" in blob
