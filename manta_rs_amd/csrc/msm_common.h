// Shared by the parts of msm_impl.h: includes and the register-cap attributes of the tail kernels.
#pragma once
#include "ec_dev.h"
#include "engine.h"
#include "fpr_dev.h"
#include "host_ec.h"
#include "params_gen.h"
#include "tuning.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mg {

// A/B switch: cap the registers of the tail kernels (merge, bucket reduce) so that they fit next to the two resident
// wavefronts of an accumulate kernel of a neighbouring MSM (160 VGPRs each: 192 are left per SIMD lane)
#ifdef MG_TAIL_WAVES
#define MG_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(MG_TAIL_WAVES, MG_TAIL_WAVES)))
#define MG_SERIAL_ATTR MG_TAIL_ATTR
#ifdef MG_TAIL_COOP_SLIM
#define MG_TAIL_COOP_ATTR MG_TAIL_ATTR
#else
#define MG_TAIL_COOP_ATTR
#endif
#else
#define MG_TAIL_ATTR
#define MG_TAIL_COOP_ATTR
// serial_reduce holds three points (acc, sum, the loaded item): 266 VGPRs left alone = one wavefront per SIMD; capped at
// 256 (30 spilled) two fit, and the big first level (2^19 buckets at c = 20) runs at the issue rate of two wavefronts
#define MG_SERIAL_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

} // namespace mg
