// intra_kernels.hip -- intra prediction: neighbour gathering + substitution + smoothing + planar/DC/angular.
//
// Replaces intra_pred[log2-2] (hevcpred_template.c:30-357) and pred_planar / pred_dc / pred_angular
// (hevcpred_template.c:359-537).  One wavefront per transform block:
//   * lane k owns neighbour index k of both reference arrays top[-1..2N-1] / left[-1..2N-1], kept in LDS;
//   * the availability cascade of :251-286 is wave-uniform (flags come from the job), each step is one
//     parallel LDS update;
//   * the [1 2 1] / strong bilinear smoothing (:289-327) is one parallel pass;
//   * prediction: each lane produces N*N/64 samples, stored row-contiguously.
// Everything the reference derives from HEVCContext (z-scan availability, picture clipping, smoothing enables)
// arrives resolved in the job record (include/ohevc_hip.h); constrained_intra_pred streams add a side record with
// per-sample "neighbour is intra" bits and the substitution walk of :185-249 runs on one lane.
#include "common.hpp"
#include "intra_body.hpp"

namespace ohevc {

template <typename Pixel>
__global__ __launch_bounds__(64) void intra_kernel(PlaneSet planes, const ohevc_intra_job *__restrict__ jobs, int njobs, int bit_depth,
                                                   const ohevc_intra_cip *__restrict__ cips)
{
    __shared__ IntraShared sh;
    intra_body<Pixel, false>(sh, threadIdx.x, planes, jobs[blockIdx.x], bit_depth, cips);
}

}  // namespace ohevc

extern "C" int ohevc_dev_intra_batch_cip(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, int njobs,
                                         const ohevc_intra_cip *cip, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(cip) & 15) == 0, "cip records must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bit_depth == 8) hipLaunchKernelGGL((intra_kernel<uint8_t>), dim3(njobs), dim3(64), 0, st, ps, jobs, njobs, bit_depth, cip);
    else                hipLaunchKernelGGL((intra_kernel<uint16_t>), dim3(njobs), dim3(64), 0, st, ps, jobs, njobs, bit_depth, cip);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_dev_intra_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, int njobs, void *stream)
{
    return ohevc_dev_intra_batch_cip(planes, bit_depth, jobs, njobs, nullptr, stream);
}
