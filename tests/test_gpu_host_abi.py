"""The host entry points of libfnp_hip.so (fnp_host_*: rotated BEV IoU, PseudoSampler.points_in_boxes, the pseudo-label
readers, the dataloader processors) are plain CPU code, tested in the CPU suite (tests/test_host_iou.py,
tests/test_pseudo_loader.py, tests/test_data_processor.py, tests/test_preprocessed_detector.py ...).  This module runs
the SAME test functions in the `-m gpu` set too, so that the driver's GPU-box run loads and exercises that native code
as well (host-ABI tests were deselected from the driver-run set)."""
import pytest

import test_data_processor as _dp
import test_host_iou as _hi
import test_host_voxelize as _hv
import test_preprocessed_detector as _pd
import test_pseudo_loader as _pl
import test_pseudo_mixing as _pm

pytestmark = pytest.mark.gpu


def _reexport(mod, prefix):
    for name in dir(mod):
        obj = getattr(mod, name)
        if name.startswith("test_"):
            globals()[f"test_{prefix}_{name[5:]}"] = obj
        elif "fixture" in type(obj).__name__.lower() or hasattr(obj, "_pytestfixturefunction") or hasattr(obj, "_fixture_function_marker"):
            globals()[name] = obj          # module-level fixtures the re-exported tests ask for


_reexport(_hi, "host_iou")
_reexport(_pl, "pseudo_loader")
_reexport(_pm, "pseudo_mixing")
_reexport(_dp, "data_processor")
_reexport(_hv, "host_voxelize")
_reexport(_pd, "preprocessed_detector")
