"""Loaders for the reference's own code built into oracle/_ref by oracle/Makefile.
TEST INFRASTRUCTURE ONLY.  Returns None when a piece was not built (e.g. reference tree absent
and nothing prebuilt travelled with the snapshot) so tests can skip."""
import ctypes
import importlib.util
import os
import sys

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def roiaware_cpu_module():
    """The reference's pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp as a torch extension
    (only points_in_boxes_cpu is callable: the CUDA launchers are left undefined)."""
    path = os.path.join(_REF, "roiaware_pool3d_ref.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)

    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("roiaware_pool3d_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


def iou3d_gpu_lib():
    """The reference's iou3d_nms_kernel.cu compiled by hipcc for gfx950 (needs a GPU to call)."""
    path = os.path.join(_REF, "libref_iou3d_gpu.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401

    return ctypes.CDLL(path)


def pib_gpu_lib():
    """The reference's point-in-box device code (roiaware_pool3d_kernel.cu:16-36,313-336, extracted by line range at build time)
    compiled by hipcc for gfx950 behind ref_pib_gpu.hip's launcher (needs a GPU to call):
    ref_points_in_boxes(B, T, M, boxes (B,T,7), pts (B,M,3), box_idx_of_points (B,M) prefilled -1, stream)."""
    path = os.path.join(_REF, "libref_pib_gpu.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401

    lib = ctypes.CDLL(path)
    lib.ref_points_in_boxes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_points_in_boxes.restype = ctypes.c_int
    return lib
