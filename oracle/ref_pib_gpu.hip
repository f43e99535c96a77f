// Launcher of our own around the REFERENCE's point-in-box device code (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:
// lidar_to_local_coords + check_pt_in_box3d :16-36 and points_in_boxes_kernel :313-336), which oracle/Makefile extracts BY LINE RANGE
// at build time into oracle/_ref/roiaware_pib_extract.inc (git-ignored; no reference text is committed) and hipcc compiles for gfx950
// as it stands.  The rest of that file calls the CUDA runtime (cudaError_t, cudaGetLastError: its launchers) and is not built; the
// launch geometry below is the reference launcher's (:345-347: DIVUP(pts_num, 256) x batch_size blocks of 256 threads).
// Same GPU, same ocml cosf / sinf, same IEEE op sequence (-ffp-contract=off like libfnp_hip.so): the comparison with
// fnp_points_in_boxes and the Box Seeker's per-candidate counts is array_equal, face-grazing points included.
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "roiaware_pib_extract.inc"

extern "C" int ref_points_in_boxes(int batch_size, int boxes_num, int pts_num, const float *boxes, const float *pts, int *box_idx_of_points,
                                   void *stream) {
    if (batch_size <= 0 || pts_num <= 0) return 0;
    dim3 blocks((pts_num + 255) / 256, batch_size), threads(256);
    hipLaunchKernelGGL(points_in_boxes_kernel, blocks, threads, 0, (hipStream_t)stream, batch_size, boxes_num, pts_num, boxes, pts, box_idx_of_points);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
