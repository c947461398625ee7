// K6: device-wide radix sort of the (bucket key, base index) pairs. A plain library sort (hipCUB /
// rocPRIM onesweep radix sort), used as-is: it is HBM-bound, moves 8 bytes per pair and pass, and is
// a few percent of an MSM; the hand-written work is in the accumulate/merge/reduce kernels.
#include "engine.h"
#include <hipcub/hipcub.hpp>

namespace mg {

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                                       (u32 *)nullptr, (int)n, 0, 32, (hipStream_t)0);
    return bytes;
}

int sort_pairs(const u32 *keys_in, u32 *keys_out, const u32 *vals_in, u32 *vals_out, size_t n, int end_bit,
               void *tmp, size_t tmp_bytes, hipStream_t s) {
    MG_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
                                              end_bit, s));
    return MG_OK;
}

} // namespace mg
