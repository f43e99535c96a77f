// Trampolines into the REFERENCE's own kernels/launchers
// (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu), compiled unmodified by hipcc from where the
// file lies (path injected by oracle/Makefile).  No reference text is copied here.
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include REF_IOU3D_KERNEL

extern "C" {
void ref_boxes_overlap(int na, const float *a, int nb, const float *b, float *out) { boxesoverlapLauncher(na, a, nb, b, out); }
void ref_boxes_iou_bev(int na, const float *a, int nb, const float *b, float *out) { boxesioubevLauncher(na, a, nb, b, out); }
void ref_boxes_aligned_overlap(int n, const float *a, const float *b, float *out) { boxesalignedoverlapLauncher(n, a, b, out); }
void ref_nms_mask(const float *boxes, unsigned long long *mask, int n, float thresh) { nmsLauncher(boxes, mask, n, thresh); }
void ref_nms_normal_mask(const float *boxes, unsigned long long *mask, int n, float thresh) { nmsNormalLauncher(boxes, mask, n, thresh); }
}
