"""The C-ABI library loads and exports every symbol include/fnp.h declares (no compute calls)."""
import ctypes
import os
import subprocess

from findnpropagate_amd import lib


def test_header_table_and_exports_agree():
    assert os.path.exists(lib.LIB_PATH), "libfnp_hip.so not built: run __graft_entry__.build()"
    header = set(lib.header_symbols())
    table = set(lib.SIGNATURES)
    assert header == table, (sorted(header - table), sorted(table - header))
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = header - exported
    assert not missing, f"declared in fnp.h but not exported: {sorted(missing)}"


def test_library_loads_and_reports_version():
    L = lib.load()
    assert L.fnp_abi_version() == 14
    assert b"gfx950" in L.fnp_version()


def test_host_only_queries():
    L = lib.load()
    # 41 x 1440 x 1440 cells -> 11 x 360 x 360 blocks of 4x4x4
    assert L.fnp_rankgrid_num_blocks(1, 41, 1440, 1440) == 11 * 360 * 360
    # (H, W) blocks are numbered in 8x8-block patches: 45 -> 6 patches of 8 per axis
    assert L.fnp_rankgrid_num_blocks(2, 5, 180, 180) == 2 * 2 * 48 * 48
    assert L.fnp_nms_workspace_bytes(100) == 100 * 2 * 8
    assert L.fnp_rankgrid_num_summary(1, 41, 1440, 1440) == (11 * 360 * 360 + 63) // 64
    assert L.fnp_rankgrid_workspace_bytes(1, 41, 1440, 1440) > 4 * ((11 * 360 * 360 + 63) // 64)


def test_gfx950_code_object_embedded():
    out = subprocess.run(["strings", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.abspath(lib.__file__))
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f
