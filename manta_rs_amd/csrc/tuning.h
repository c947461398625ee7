// What a deployment may decide about the library, in ONE struct that crosses the C ABI (`mg_tuning`, include/mantagpu.h; mirrored
// in rust/mantagpu-sys), and what is only an A/B switch of a measurement campaign.
//
//   Tuning      graph topology, streams per proof, coalescing window, passes in flight, window widths of the key tables, table budget,
//               queue placement. Process-wide defaults come from the environment ONCE, through the table in runtime.cpp (the only
//               place the shipped library reads MANTA_* variables: 14 names, listed by mg_tuning_env_names); a host sets them
//               through mg_set_tuning before it creates contexts, or per context through mg_ctx_opts.tuning. None changes a
//               result: tests/test_gpu_profiles.py proves every field and every variable leaves proof bytes unchanged.
//   ab_knob     every other knob of rounds 1-6 was measured and closed (DESIGN.md section 7 names the file that holds each result):
//               the shipped library compiles its default in. A unit rebuilt with -DMG_DIAG (tools/build_variant.sh <tag> -DMG_DIAG
//               <unit>) reads it from the environment again, which is how the A/B tools re-measure one.
#pragma once
#include <cstdint>
#include <cstdlib>

namespace mg {

struct Tuning { // field for field `mg_tuning`
    uint32_t struct_size;
    int32_t graph_mode;           // 1 single (default: two graphs per pass), 2 split (six single-stream graphs), 0 off (eager launches)
    int32_t graph_mode_batch;     // the same for passes of >= 4 proofs; -1 = as graph_mode
    int32_t prove_streams;        // streams of a forked pass: 3, 4, 5 or 6 (default)
    int32_t linear_chains;        // single proofs as three linear graphs: 0 never, 1 a lone proof, 2 also beside other passes, 3 (default) + chain choice
    int32_t coalesce_inflight;    // passes of coalesced concurrent single calls on the GPU: 0 = no coalescing .. 4, default 2
    int32_t coalesce_gather_us;   // how long the leader of a coalesced pass waits for the callers of the pass that just ended (default 100)
    int32_t batch_inflight;       // passes of one mg_groth16_prove_batch call in flight (default 3)
    int32_t queue_aware;          // 1 (default): single-proof slots get streams on measured hardware queues (queues.hip)
    int32_t msm_dedicated_queues; // stand-alone MSMs on streams with a hardware queue of their own: 1 (default) while no context is alive, 2 always, 0 never
    int32_t window_bits_narrow;   // key tables, 0 = the library's choice: latency tables of a / b_g1 / l (default 8 at manta-pay sizes)
    int32_t window_bits_wide;     //   batched-pass tables (default 12 from 2^15 scalars on, else 11)
    int32_t window_bits_h;        //   the h query (default 12 / log2(D) - 2)
    int32_t window_bits_g2;       //   b_g2 (default 6)
    int64_t full_table_bytes;     // HBM budget of a context's full tables; -1 = the default (a tenth of the device), 0 = none
};
constexpr int GRAPH_MODE_OFF = 0, GRAPH_MODE_SINGLE = 1, GRAPH_MODE_SPLIT = 2;

Tuning tuning_defaults();          // compiled-in defaults, no environment
const Tuning &tuning();            // the process-wide values in force (environment applied once, then mg_set_tuning)
int set_tuning(const Tuning &t);   // validates; MG_OK / MG_ERR_ARG
int normalize_tuning(Tuning &t);   // clamps nothing, rejects out-of-range fields: MG_OK / MG_ERR_ARG
// the environment variables the shipped library reads (tuning table + MANTA_RCCL_LIB), NULL-terminated
const char *const *tuning_env_names();

#ifdef MG_DIAG
inline int ab_knob(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
inline const char *ab_knob_str(const char *name, const char *dflt) {
    const char *e = std::getenv(name);
    return e ? e : dflt;
}
#else
constexpr int ab_knob(const char *, int dflt) { return dflt; }
constexpr const char *ab_knob_str(const char *, const char *dflt) { return dflt; }
#endif

} // namespace mg
