// oc_step_kernel.h -- the ahead-of-time (generic) build of the step + encode kernel: every layout parameter a
// runtime field of StepArgs.  The per-engine / per-group specialised builds are generated at run time from the
// same two headers (overcooked_engine.hip: jit_source()).
#pragma once

#include "oc_step_common.h"

#define OC_NS generic
#include "oc_step_body.h"
#undef OC_NS

namespace oc_dev {

template <bool STEP, bool ACT64, bool VEC16, int QN>
__global__ __launch_bounds__(kWave) void oc_step_kernel(const StepArgs a) {
    generic::step_body<STEP, ACT64, VEC16, QN>(a, blockIdx.x, gridDim.x);
}
template <bool ACT64, bool VEC16, int QN>
__global__ __launch_bounds__(kWave) void oc_rollout_kernel(const StepArgs a) {
    generic::step_body<true, ACT64, VEC16, QN, SM_PLAIN, true>(a, blockIdx.x, gridDim.x);
}
template <bool VEC16, int QN>
__global__ __launch_bounds__(kWave) void oc_observe_batch_kernel(const StepArgs a) {
    generic::step_body<false, false, VEC16, QN, SM_PLAIN, false, true>(a, blockIdx.x, gridDim.x);
}

}  // namespace oc_dev
