"""ctypes binding of libfnp_hip.so — the C ABI declared in include/fnp.h.

This is the only way product code reaches the GPU kernels.  There is NO CPU fallback: if the
shared library is missing the import of any operator fails loudly (build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C findnpropagate_amd/csrc``).

PyTorch is used for device memory and streams only: tensors are passed as raw device pointers
and every call is enqueued on ``torch.cuda.current_stream()``.
"""
import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FNP_LIB_PATH: development override used by tools/ to A/B kernel builds; never set in production
LIB_PATH = os.environ.get("FNP_LIB_PATH") or os.path.join(_HERE, "libfnp_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fnp.h")

FNP_F32 = 0
FNP_BF16 = 1
FNP_F16 = 2

_ERRORS = {-1: "FNP_ERR_ARG", -2: "FNP_ERR_LAUNCH", -3: "FNP_ERR_HIP", -4: "FNP_ERR_WORKSPACE"}


class FnpError(RuntimeError):
    pass


class VoxelCfg(ctypes.Structure):
    """struct fnp_voxel_cfg (include/fnp.h)."""

    _fields_ = [
        ("range_min", c_float * 3),
        ("voxel_size", c_float * 3),
        ("grid", c_int * 3),
        ("num_features", c_int),
        ("max_points", c_int),
        ("max_voxels", c_int),
    ]


class RankGridC(ctypes.Structure):
    """struct fnp_rankgrid (include/fnp.h)."""

    _fields_ = [
        ("B", c_int), ("D", c_int), ("H", c_int), ("W", c_int),
        ("bits", c_void_p), ("base", c_void_p), ("summary", c_void_p), ("perm", c_void_p), ("counters", c_void_p),
    ]


class SeekerParams(ctypes.Structure):
    """struct fnp_seeker_params (include/fnp.h)."""

    _fields_ = [
        ("lq", c_float), ("uq", c_float), ("cq", c_float),
        ("iou_w", c_float), ("dst_w", c_float), ("dns_w", c_float),
        ("min_cam_iou", c_float), ("max_dist", c_float),
        ("num_mags", c_int), ("num_rotations", c_int), ("num_sizes", c_int),
        ("topk", c_int), ("clamp_bottom", c_int), ("image_h", c_int), ("image_w", c_int),
        ("point_stride", c_int), ("xyz_offset", c_int),
        ("has_img_aug", c_int), ("mult", c_int), ("ego_w", c_float),
        ("nms_normal", c_float), ("search_depth", c_float), ("occl_w", c_float),
        ("occl_mult", c_int), ("multicam", c_int), ("count_only", c_int), ("num_frustums", c_int),
        ("npts_all", ctypes.c_void_p), ("rand_noise", ctypes.c_void_p),
    ]


class ConvGeom(ctypes.Structure):
    """struct fnp_conv_geom (include/fnp.h)."""

    _fields_ = [
        ("ksize", c_int * 3),
        ("stride", c_int * 3),
        ("padding", c_int * 3),
        ("in_shape", c_int * 3),
        ("out_shape", c_int * 3),
    ]


P = c_void_p  # every device pointer crosses the ABI as a plain address

# name -> (restype, argtypes).  Mirrors include/fnp.h one to one; tests/test_abi.py checks that
# the header, this table and the exported symbols agree.
SIGNATURES = {
    "fnp_version": (c_char_p, []),
    "fnp_abi_version": (c_int, []),
    "fnp_points_in_boxes": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "fnp_points_in_boxes_count": (c_int, [P, P, P, c_int, c_int, P]),
    "fnp_points_in_boxes_dense": (c_int, [P, P, P, c_int, c_int, P]),
    "fnp_boxes_overlap_bev": (c_int, [P, c_int, P, c_int, P, P]),
    "fnp_boxes_iou_bev": (c_int, [P, c_int, P, c_int, P, P]),
    "fnp_boxes_aligned_overlap_bev": (c_int, [P, P, c_int, P, P]),
    "fnp_boxes_iou3d": (c_int, [P, c_int, P, c_int, P, P]),
    "fnp_boxes_aligned_iou3d": (c_int, [P, P, c_int, P, P]),
    "fnp_recall_counters": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int, P, c_int, c_uint, c_uint, P, P]),
    "fnp_seeker_pack_records": (c_int, [P, P, P, c_int, P, c_int, c_int, P, P]),
    "fnp_host_points_in_boxes_frame": (c_int, [P, c_int, c_int, P, c_int, P, P]),
    "fnp_host_boxes_iou_bev": (c_int, [P, c_int, P, c_int, P]),
    "fnp_host_boxes_aligned_iou_bev": (c_int, [P, P, c_int, P]),
    "fnp_nms_workspace_bytes": (c_int64, [c_int]),
    "fnp_nms_rotated": (c_int, [P, c_int, c_float, P, P, P, P]),
    "fnp_nms_normal": (c_int, [P, c_int, c_float, P, P, P, P]),
    "fnp_nms_batched_workspace_bytes": (c_int64, [c_int, c_int]),
    "fnp_nms_batched": (c_int, [P, P, c_int, c_int, c_float, c_int, P, P, P, P]),
    "fnp_rankgrid_num_blocks": (c_int64, [c_int, c_int, c_int, c_int]),
    "fnp_rankgrid_num_summary": (c_int64, [c_int, c_int, c_int, c_int]),
    "fnp_rankgrid_counter_words": (c_int64, [c_int, c_int, c_int, c_int]),
    "fnp_rankgrid_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "fnp_rankgrid_build": (c_int, [P, P, c_int, POINTER(RankGridC), P, c_int64, P]),
    "fnp_rankgrid_clear": (c_int, [P, P, c_int, POINTER(RankGridC), P]),
    "fnp_rankgrid_clear_multi": (c_int, [c_int, P, P, P, P, P]),
    "fnp_rankgrid_clear_summary": (c_int, [c_int, c_void_p, c_void_p]),
    "fnp_voxelize_workspace_bytes": (c_int64, [c_int64, POINTER(VoxelCfg), POINTER(RankGridC)]),
    "fnp_stage_points": (c_int, [P, c_int64, c_int64, c_int, c_float, P, P, c_int, P, P]),
    "fnp_voxelize": (c_int, [P, c_int, P, POINTER(VoxelCfg), POINTER(RankGridC), P, c_int64,
                             P, P, P, P, P, P, c_int, P]),
    "fnp_host_voxelize": (c_int, [P, c_int, POINTER(VoxelCfg), P, P, P, c_int]),
    "fnp_rulebook_subm": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), P, P]),
    "fnp_rulebook_strided": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), POINTER(RankGridC),
                                     P, P, c_int, P, P, c_int64, P]),
    "fnp_spconv_forward": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, c_int, P, c_int,
                                   P, P, P, c_int, c_int, c_int, c_int, P]),
    "fnp_spconv_forward_strided": (c_int, [P, c_int, c_int, P, POINTER(RankGridC), POINTER(ConvGeom), P, P, c_int, P, c_int,
                                           P, P, c_int, c_int, c_int, P]),
    "fnp_spconv_tiled_aborts": (c_int, []),
    "fnp_gather_counts": (c_int, [P, c_int, c_uint, P, P]),
    "fnp_gather_counts_host": (c_int, [P, c_int, c_uint, P, P, P, P]),
    "fnp_spconv_tiled_aborts_copy": (c_int, [P, P]),
    "fnp_debug_tile_hold": (c_int, [c_int]),
    "fnp_tile_rulebook_bytes": (c_int64, [c_int, c_int]),
    "fnp_tile_rulebook_build": (c_int, [P, c_int, c_int, P, c_int, c_int, P, P]),
    "fnp_rulebook_subm_tiled": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), P, c_int, P, P]),
    "fnp_rulebook_subm_tiled_lean": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), P, c_int, P, POINTER(RankGridC), POINTER(ConvGeom), P, P]),
    "fnp_rulebook_strided_premarked": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), POINTER(RankGridC),
                                               P, P, c_int, P, P, c_int64, P]),
    "fnp_spconv_forward_tiled": (c_int, [P, c_int, c_int, P, P, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    "fnp_split_bf16": (c_int, [P, P, c_int, c_int, P, P, P]),
    "fnp_split_bf16_add": (c_int, [P, P, c_int, P, c_int, c_int, P, P, P, P]),
    "fnp_spconv_forward_split": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "fnp_spconv_forward_sorted_split": (c_int, [P, c_int, c_int, P, P, c_int, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fnp_spconv_forward_tiled_split": (c_int, [P, c_int, c_int, P, P, P, c_int, P, c_int, P, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fnp_ell_bytes": (c_int64, [c_int, c_int]),
    "fnp_rulebook_ell": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), P, c_int, P, c_int, P, P]),
    "fnp_spconv_forward_ell": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, P, P, P, c_int, c_int, c_int, P]),
    "fnp_spconv_forward_ell_mfma": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, P, P, P, c_int, c_int, c_int, P]),
    "fnp_classsort_workspace_bytes": (c_int64, [c_int]),
    "fnp_rulebook_subm_masked": (c_int, [P, P, c_int, POINTER(ConvGeom), POINTER(RankGridC), P, P, POINTER(RankGridC), POINTER(ConvGeom), P]),
    "fnp_rulebook_classsort": (c_int, [P, c_int, c_int, P, P, c_int, c_int, c_int, P, P, P, c_int64, P]),
    "fnp_rulebook_classsort_f32": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "fnp_spconv_forward_f32_sorted": (c_int, [P, c_int, P, P, c_int, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "fnp_spconv_forward_sorted": (c_int, [P, c_int, c_int, P, P, c_int, P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    "fnp_boxseeker_workspace_bytes": (c_int64, [c_int, c_int]),
    "fnp_seeker_prepare_matrices": (c_int, [P, P, P, P, P, c_int, P, P, P]),
    "fnp_host_enumerate_frustums": (c_int, [P, P, P, P, P, c_int, c_int, POINTER(c_int), c_int, c_float, c_float, P, c_int]),
    "fnp_boxseeker": (c_int, [P, P, c_int, c_int, POINTER(SeekerParams), P, P, P, c_int, P, P, P, P, c_int64,
                              P, P, P, P, P, P, P, P, P, P, P]),
    "fnp_rulebook_transpose": (c_int, [P, c_int, c_int, P, c_int, P, c_int, P]),
    "fnp_spconv_wgrad_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "fnp_spconv_wgrad": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, c_int, P, c_int, c_int, c_int, P, c_int64, P]),
    "fnp_pack_weight": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "fnp_pack_weight_multi": (c_int, [c_int, P, P, P, P, c_int, P, P, P, P]),
    "fnp_rulebook_pairs_workspace_bytes": (c_int64, [c_int, c_int]),
    "fnp_rulebook_pairs": (c_int, [P, c_int, c_int, P, c_int, P, P, c_int, P, P, c_int64, P]),
    "fnp_spconv_wgrad_pairs": (c_int, [P, c_int, P, c_int, P, P, P, c_int, c_int, P, c_int, P, c_int, c_int, c_int, P, c_int64, P]),
    "fnp_bn_workspace_bytes": (c_int64, [c_int]),
    "fnp_bn_train_forward": (c_int, [P, c_int, P, c_int, c_int, P, P, P, P, c_float, c_float, P, c_int, P, P, P, P, P, c_int64, P]),
    "fnp_bn_train_backward": (c_int, [P, P, P, c_int, P, c_int, c_int, P, P, P, c_int, P, P, P, P, P, c_int64, P]),
    "fnp_clipcrop_plan": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fnp_clipcrop_sample": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, P, P]),
    "fnp_sparse_to_dense_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "fnp_sparse_to_dense": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64, P]),
    "fnp_sparse_to_dense_fill": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P]),
}


def header_symbols():
    """Function names declared in include/fnp.h (used by the ABI test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fnp_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load():
    """dlopen libfnp_hip.so and bind every entry point.  Raises FnpError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FnpError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for this path."
        )
    # torch ships its own libamdhip64 (same SONAME); importing it first makes this library bind
    # to the runtime torch already loaded, so both share one HIP context and stream table.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise FnpError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("FNP_TRACE"):
        lib = _Traced(lib)     # development: name every call and synchronise behind it (attributes a GPU fault to its launch)
    _lib = lib
    return lib


class _Traced:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("fnp_") or "workspace" in name or name.startswith("fnp_host") or name.startswith("fnp_rankgrid_num"):
            return fn

        def call(*a):
            import sys
            import torch
            capturing = torch.cuda.is_current_stream_capturing()
            print(f"[fnp] {name}{' (capture)' if capturing else ''} {[x for x in a if isinstance(x, int) and abs(x) < 1 << 31]}", file=sys.stderr, flush=True)
            rc = fn(*a)
            if not capturing and os.environ.get("FNP_TRACE") != "2":     # FNP_TRACE=2: names only, no synchronisation
                torch.cuda.synchronize()
            return rc
        return call


def check(code, what):
    if code != 0:
        raise FnpError(f"{what} failed: {_ERRORS.get(code, code)}")


def ptr(t):
    """Device (or host) address of a contiguous torch tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "fnp ABI takes contiguous buffers"
    return t.data_ptr()


_raw_stream = None


def stream():
    """hipStream_t of torch's current stream, as an integer handle for the ABI."""
    global _raw_stream
    if _raw_stream is None:
        import torch

        raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and dev is not None:   # (0.2 us instead of 2.8 for torch.cuda.current_stream().cuda_stream: one call per launch)
            _raw_stream = lambda: raw(dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _raw_stream()


def dtype_code(t):
    import torch

    if t.dtype == torch.float32:
        return FNP_F32
    if t.dtype == torch.bfloat16:
        return FNP_BF16
    if t.dtype == torch.float16:
        return FNP_F16
    raise FnpError(f"unsupported feature dtype {t.dtype}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FnpError("fnp kernels take device tensors; got a CPU tensor (there is no CPU path)")
