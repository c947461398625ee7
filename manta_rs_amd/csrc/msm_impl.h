// Pippenger variable-base MSM on gfx950 -- kernels and host orchestration, templated on the curve
// and group. Instantiated once per (curve, group) in msm_<curve>_<group>.hip.
//
// Replaces ark-ec ^0.3.0 `VariableBaseMSM::multi_scalar_mul` (4x G1 + 1x G2 per proof, SURVEY.md rows
// a-7/a-8; reached from manta-crypto/src/arkworks/groth16.rs:597). The result is the same group
// element; the schedule is GPU-native and differs from arkworks' on purpose:
//
//   K5 digits      scalar -> W signed c-bit digits -> (bucket key, base index|sign) pairs, coalesced
//   K6 sort        one device-wide radix sort of the n*W pairs by bucket key (sort.hip)
//   K7a chunks     every lane walks L consecutive sorted pairs, mixed-adding gathered bases into an
//                  XYZZ accumulator kept in VGPRs; bucket runs that start and end inside the chunk are
//                  stored straight to their bucket, the chunk's first/last run become "partials"
//   K7b merge      partials (still sorted by key) are combined by a wavefront-wide segmented scan
//                  (shuffles, no LDS, no atomics), 64 -> 2 per wave and level, until one wave is left
//   K8 reduce      sum_k (k+1) B_k per bucket window as wavefront suffix-scan + reduction per 64-bucket
//                  tile, two levels; <= a few dozen points per window go to the host
//   K9 (host)      fold those points, Horner over windows (none when the bases carry precomputed
//                  2^(c w) multiples: all windows then share ONE bucket set and no doubling is left)
//
// Load balance never depends on the scalar distribution: work is split by sorted position, not by
// bucket, so a witness that is 25 % ones (one giant bucket) costs the same as uniform scalars.
// No global atomics on points; every bucket is written exactly once; results are deterministic.
//
// Layout (round 6: the 2 400-line header split along its seams):
//   msm_common.h      includes, register-cap attributes shared by the kernels
//   msm_digits.h      K5   digits_kernel
//   msm_accumulate.h  K7a  accumulate_chunks / accumulate_single
//   msm_reduce.h      K7b-K9 merge_partials*, tile_reduce*, serial_reduce*, reduce_level1*, fold_windows
//   msm_tables.h      bases_to_internal, window / full tables, fixed-base, group NTT, element-wise kernels
//   msm_engine.h      GroupEngineT: base sets, plan, launch, finish (host)
#pragma once
#include "msm_engine.h"
