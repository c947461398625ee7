// Host-side runtime interfaces of the MI355X Groth16 hot path (one process per GPU; every object
// below lives on the current HIP device). Curve/group-specific code sits behind GroupEngine so the
// C ABI (include/mantagpu.h) can dispatch on (curve, group) at run time.
#pragma once
#include <hip/hip_runtime.h>
#include <shared_mutex>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace mg {

typedef uint32_t u32;
typedef uint64_t u64;

// Wave issue priority of everything that is NOT the bucket-accumulate kernel. A SIMD's arbiter serves the oldest wavefront
// first; next to the two resident accumulate wavefronts of a neighbouring MSM (VALU busy 91 %) a younger wavefront of a
// sort / merge / reduce kernel got one issue slot in ten -- traced on MI355X with three MSMs in flight: radix_scatter 1 232 us
// instead of 110, tile_reduce_coop 1 842 instead of 110. These kernels are short chains with little total work, so they go
// first (s_setprio 3) and the accumulate kernel of the other MSM fills every slot they leave.
// (Measured and not adopted, profiles/r03_priority_ab.txt: the witness-map kernels at the same priority -- a sequential proof got
// 4 % slower, they took issue slots from the G2 chain, the batched stream did not move; the G2 MSM's kernels one level above the G1
// ones -- no difference beyond run-to-run noise.)
#ifdef MG_NO_PRIO
#define MG_PRIO_HIGH() ((void)0)
#else
#define MG_PRIO_HIGH() __builtin_amdgcn_s_setprio(3)
#endif
#define MG_PRIO_FOR(F) MG_PRIO_HIGH()

enum { MG_OK = 0, MG_ERR_ARG = 1, MG_ERR_HIP = 2, MG_ERR_OOM = 3, MG_ERR_DOMAIN = 4, MG_ERR_STATE = 5 };

#define MG_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            mg::set_last_hip_error(e_, #expr, __FILE__, __LINE__);                                                \
            return (e_ == hipErrorOutOfMemory) ? mg::MG_ERR_OOM : mg::MG_ERR_HIP;                                 \
        }                                                                                                         \
    } while (0)
void set_last_hip_error(hipError_t e, const char *expr, const char *file, int line);
void set_last_error_text(const char *text); // detail for mg_last_error() of a failure that is not a HIP status
// When on, every MSM brackets its accumulate kernel with HIP events on the launch stream (bench.py's
// roofline leg); off by default.
void set_kernel_timing(bool on);
bool kernel_timing();
void set_last_accumulate_ms(float ms);
float last_accumulate_ms();
void set_last_accumulate_mhz(float mhz); // shader clock of the same launch: s_memtime ticks of its first wavefront per wall-clock second
float last_accumulate_mhz();
// with kernel timing on: the last mg_ntt / mg_ntt_device of this thread -- [0] whole call on the device, [1] conversion in,
// [2] butterfly passes, [3] conversion out (ms); and the last single proof of this thread, run with eager launches --
// [0] upload of z, [1] witness map, [2..6] MSM a, b_g1, b_g2, l, h (each on its own stream), [7] part A (everything but
// the G2 MSM) from upload to join, [8] G2 MSM from upload to its end, [9] host assembly after the GPU (ms)
void set_last_ntt_ms(const float v[4]);
void get_last_ntt_ms(float v[4]);
void set_last_pass_host_ms(const float v[3]); // host side of the calling thread's last pass: enqueue, wait for the GPU, assembly
void get_last_pass_host_ms(float v[3]);
void set_last_prove_ms(const float v[10]);
void get_last_prove_ms(float v[10]);
// process-lifetime pool of non-blocking streams for proof slots (returned, never destroyed -- runtime.cpp)
constexpr int MAX_DEVICES = 16;
int current_device(); // hipGetDevice, clamped to the engine tables
// Stream capture against the rest of the process (round 5, found by tools/soak.py): while ANY thread captures a proof slot's graphs,
// another thread that creates or destroys a context / base set / verifying context (synchronous copies, kernels on the default
// stream, hipDeviceSynchronize, hipFree) makes the runtime invalidate the capture -- every stream that had joined it then answers
// hipErrorStreamCaptureInvalidated (901) to every later call and the slot is lost. Captures hold this lock exclusively (about a
// millisecond, once per slot), the heavy operations hold it shared: they still run beside each other and beside proofs that replay.
std::shared_mutex &capture_mutex();
struct HeavyOp { // the shared side, re-entrant per thread (a failed creation destroys what it built; a context owns its peers)
    HeavyOp();
    ~HeavyOp();
    HeavyOp(const HeavyOp &) = delete;
    HeavyOp &operator=(const HeavyOp &) = delete;
};
// The library puts NOTHING on the NULL stream (round 6). Key upload, table precompute, domain tables, verifying keys and the
// synchronous copies of the C ABI used the default stream and hipDeviceSynchronize; the NULL stream is an ordinary stream on one
// of the runtime's shared hardware queues, every BLOCKING stream of the process orders itself against it (the stand-alone MSMs'
// dedicated-queue streams are blocking), and whatever shares its queue stalls behind that ordering: tools/soak.py with stand-alone
// MSMs beside proofs and a context recycled every 3 s ran at 360 proofs/s instead of 3 400. Setup work now runs on a pooled
// NON-blocking stream owned by the calling thread for the current device (returned to the pool when the thread ends); the
// "synchronous" copies are asynchronous copies on it followed by a wait for that stream alone.
hipStream_t setup_stream();
inline hipError_t memcpy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    hipStream_t s = setup_stream();
    if (!s) return hipErrorOutOfMemory;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e == hipSuccess ? hipStreamSynchronize(s) : e;
}
inline hipError_t setup_sync() { // everything this thread enqueued on its setup stream has completed
    hipStream_t s = setup_stream();
    return s ? hipStreamSynchronize(s) : hipErrorOutOfMemory;
}
hipStream_t stream_pool_get();
void stream_pool_put(hipStream_t s);
hipStream_t stream_pool_get_normal(); // normal-priority streams: pooled and never destroyed either (runtime.cpp)
void stream_pool_put_normal(hipStream_t s);
// Proving / verifying contexts alive in this process. A stand-alone MSM takes its dedicated-queue stream only while there is NONE
// (tuning msm_dedicated_queues = 1, the default; 2 = always, 0 = never): beside proof passes the dedicated queues cost far more than
// they give -- tools/soak.py with stand-alone MSMs next to six proving threads: 3 380 proofs/s on ordinary streams, 640 with the MSMs
// on dedicated queues, also with graphs off and with nothing on the NULL stream (profiles/r06_soak.txt; the runtime's handling of
// CU-masked HSA queues next to the pooled ones -- not identified further). The pipeline of MSMs of a process that only does MSMs
// (BASELINE configs[1]) is where they pay: 390-398 Mscalar/s for every stream-creation order against 317-394.
struct GraphClient { // a member of every proving / verifying context
    GraphClient();
    ~GraphClient();
    GraphClient(const GraphClient &) = delete;
    GraphClient &operator=(const GraphClient &) = delete;
};
int graph_clients_alive();
hipStream_t stream_pool_get_dedicated(); // a stream on a hardware queue of its own (MsmWorkspace::solo); nullptr: off / refused
void stream_pool_put_dedicated(hipStream_t s);
// Queue-aware streams of a single-proof slot (queues.hip, runtime.cpp): three streams on three DIFFERENT hardware queues, chosen so
// that slots with neighbouring ids share as few queues as the hardware has (set i: high-priority classes 2i and 2i+1 for the
// witness map + h chain and the G2 chain, normal-priority class i -- or high-priority class 2i+2 -- for the combined MSM).
struct StreamSet {
    hipStream_t main = nullptr, g2 = nullptr, z3 = nullptr;
    int id = -1, dev = 0;
    bool z3_high = false;
};
bool stream_sets_ready();                            // probes the current device's queues on first use; false: disabled / failed
int stream_queue_counts(int *normal, int *high);     // hardware queues found per priority level (0: not probed)
bool stream_set_acquire(StreamSet &s, bool z3_high); // false: queues unknown -- use the plain pools
void stream_set_release(StreamSet &s);
int streams_share_queue(hipStream_t a, hipStream_t b, unsigned *mem, unsigned *token_counter); // queues.hip
const char *last_error_string();

// makes `dev` the current device for a scope and restores the caller's on the way out: a host thread that drives several
// GPUs must not find its device changed by a call into the library
struct DeviceScope {
    int prev = 0;
    bool switched = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        if (switched) hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// restores the caller's current device when a call that visits other devices returns
struct DeviceGuard {
    int prev = 0;
    DeviceGuard() { hipGetDevice(&prev); }
    ~DeviceGuard() { hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T *as() const { return (T *)p; }
};

enum { SCALARS_CANONICAL = 0, SCALARS_MONT = 1, SCALARS_WORK = 2 };

// Pippenger plan. Signed digits of c bits, B = 2^(c-1) buckets per bucket-window.
struct MsmPlan {
    int c = 0;       // window bits
    int W = 0;       // windows = ceil(scalar_bits / c): digits_kernel folds k > r / 2 to r - k
    u32 B = 0;       // buckets per window
    int Wb = 0;      // bucket windows: W (plain) or 1 (bases precomputed for every window)
    bool precomp = false;
    bool full = false; // every multiple 1..B of every window tabulated: digits address table entries, no buckets
    u32 L = 0;       // sorted entries per thread in the chunk-accumulate kernel
};

// A static set of bases resident in HBM (a proving-key query, or a caller-registered vector).
struct BaseSet {
    int curve = 0, group = 1;
    int device = 0;       // the HIP device the points live on
    size_t n = 0;         // points stored (after dropping infinity entries when compacted)
    size_t n_orig = 0;    // logical length: scalars are indexed 0 .. n_orig-1
    u32 *d_map = nullptr; // compacted sets: stored point i belongs to scalar d_map[i]; nullptr = identity
    u32 *d_pts = nullptr; // affine AoS; with precompute: W tables of n points, table w = 2^(c w) * P
    int pre_c = 0, pre_W = 0; // 0 = no precompute
    bool full = false;        // with precompute: the B = 2^(c-1) multiples m 2^(c w) P of every window too, entry ((w n + i) B + m - 1)
    // Round 4: a set may be the CONCATENATION of n_sets queries of set_len entries each that are multiplied by the SAME scalar
    // vector (a_query | b_g1_query | l_query of a proving key): entry j belongs to query j / set_len and to scalar j % set_len,
    // every query gets its own bucket keys, and ONE pass of the pipeline -- one digit kernel, one accumulate launch, one chain of
    // merge levels -- leaves n_sets results. n_sets = 1: an ordinary set.
    u32 n_sets = 1;
    size_t set_len = 0;
    // stored points of query q = [set_first[q], set_first[q + 1]) (infinity entries dropped, order kept): with full tables the
    // digit kernel runs once per query, in stream order, and the pairs come out grouped by query -- no sort (msm_launch)
    static constexpr u32 MAX_SETS = 4;
    u32 set_first[MAX_SETS + 1] = {};
    size_t bytes = 0;
};

// Per-call scratch for one MSM (device + pinned host staging). Pooled per engine.
struct MsmWorkspace {
    DevBuf keys_in, keys_out, vals_in, vals_out, sort_tmp, buckets, pkeys[2], ppts[2], redA, redS, misc;
    DevBuf count; // number of (key, value) pairs the digit kernel produced (zero digits are compacted away)
    // work-efficient front levels of the bucket reduce (msm_impl.h serial_reduce): per-lane (A, S) arrays and the
    // temporaries of the plain sums; `extra` = one staged point per (front level, window segment)
    DevBuf front, extra;
    static constexpr int MAX_EXTRA = 8;
    u32 tail_shift = 0, n_extra = 0, extra_shift[MAX_EXTRA] = {}; // window = 2^tail_shift * tail + sum_e 2^shift_e * extra_e
    size_t extra_off_pts = 0;                                       // where the extras start in h_stage (points)
    const u32 *d_tail = nullptr; // device copy of what msm_launch staged for the host fold (arkworks-format XYZZ points)
    DevBuf folded;               // msm_fold_device: one arkworks-format XYZZ point per batch member
    // kernel timing: (s_memtime ticks, wall-clock ticks) of the accumulate kernel's first wavefront, written by the kernel straight to
    // page-locked HOST memory and read after the `done` event -- no copy on the NULL stream, which would order itself against every
    // blocking stream of the process (the stand-alone MSMs' dedicated-queue streams are blocking: see `solo`)
    unsigned long long *h_clk = nullptr;
    DevBuf scratch;              // caller-side scalars uploaded for one launch (the verifier's small MSMs)
    void *h_stage = nullptr; // pinned
    size_t h_stage_cap = 0;
    hipStream_t stream = nullptr;
    hipStream_t run_on = nullptr; // when set, msm_launch enqueues on this stream instead of the workspace's own
    // Round 6 -- stand-alone MSMs (mg_msm_launch: no proof slot, no capture) run on a stream with a HARDWARE QUEUE OF ITS OWN. The HIP
    // runtime multiplexes ordinary streams onto four hardware queues per priority level in creation order, kernels of streams that
    // share a queue run one behind the other, and which of a process's streams shared one decided the pipelined 2^20 MSM rate:
    // 317-394 Mscalar/s by the number of streams another library had created first (profiles/r05_hw_queues.txt (6)). A stream
    // created through hipExtStreamCreateWithCUMask gets an HSA queue of its own, whatever its mask -- the mask here names every CU
    // (profiles/r06_pipeline_phase.txt: 379-387 for every creation order). Such a stream is a BLOCKING stream (the extension takes
    // no flags): it orders itself against NULL-stream work of the process; the library puts nothing on the NULL stream between an
    // MSM's launch and its finish. Owned by the workspace, pooled with it, never destroyed.
    hipStream_t solo = nullptr;
    bool use_solo = false; // set by the stand-alone entry points for the lifetime of one job
    // stand-alone MSMs: the plain sums of the bucket reduce's front levels run here, beside the weighted chain
    hipStream_t side_stream = nullptr; // the ENGINE's side stream once this workspace has used it (not owned)
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr; // optional timing of the dominant (accumulate) kernel
    bool timed = false;
    bool in_graph_slot = false; // owned by a proof slot (prover_passes.h): its launches never use the engine's side stream
    bool capturing = false; // msm_launch is being stream-captured: enqueue kernels and copies only, no event records
    // notify: after the staged result a 4-byte token is copied to *h_flag (pinned), so that a host that zeroed it before the
    // launch can see THIS chain end without a stream or event wait (a chain inside a captured multi-branch graph has neither)
    bool notify = false;
    u32 *h_flag = nullptr; // pinned, owned
    u32 *d_token = nullptr; // device word holding 1, owned
    float accumulate_ms = 0.f;
    // host-side description of what was staged (filled by msm_launch, consumed by msm_finish)
    MsmPlan plan;
    u32 T1 = 0, nP = 0, batch = 1;
    int pending = 0;
    int device = 0;
    ~MsmWorkspace();
};

// the stream an MSM of this workspace is enqueued on
inline hipStream_t msm_stream_of(const MsmWorkspace *ws) {
    return ws->run_on ? ws->run_on : (ws->use_solo && ws->solo ? ws->solo : ws->stream);
}

// Opaque host point (XYZZ, 64-bit limbs) big enough for G2/BLS12-381.
struct HostPoint {
    u64 w[4 * 12];
};

class GroupEngine {
  public:
    virtual ~GroupEngine() {}
    virtual int curve() const = 0;
    virtual int group() const = 0;
    virtual int affine_words() const = 0; // u32 per affine point
    virtual int xyzz_words() const = 0;
    virtual int scalar_bits() const = 0;
    virtual int base_record_bytes() const = 0; // HBM bytes of one stored base point (internal affine record, padded)

    // bases: affine Montgomery AoS (host or device pointer). precompute_c > 0 builds 2^(c w) tables.
    // drop_infinity (host sources only): points at infinity are removed from the stored set (proving-key
    // queries are full of them: every variable absent from B has b_g1_query = b_g2_query = infinity) and
    // never reach the sort or the accumulate kernel; scalars stay indexed by the original positions.
    virtual int bases_create(const u32 *pts, size_t n, bool src_on_device, int precompute_c, BaseSet **out,
                             bool drop_infinity = false, u32 n_sets = 1) = 0;
    virtual void bases_destroy(BaseSet *) = 0;

    virtual MsmPlan plan_for(const BaseSet *bs, size_t n, int c_override, u32 batch = 1) const = 0;
    // Enqueue the whole MSM on ws->stream: scalars are device-resident (canonical, or Montgomery if
    // scalars_mont), n <= bs->n. Result is staged to pinned memory; call msm_finish to fold it.
    // batch > 1: `batch` independent scalar vectors (vector q starts scalar_stride_words u32 after vector
    // q-1) against the SAME bases in one pass of the pipeline -- every (vector, window) pair is its own
    // bucket segment, so the kernels run once over batch x the entries (a batch of proofs of one circuit).
    // sparse: the scalars are expected to have many zero digits (a witness: 40 % zeros, 25 % ones) -- the digit
    // kernel then compacts the zero digits away before the sort (two passes over the digits and one atomic per
    // wavefront: ~3 % slower on uniform scalars, up to 18 % faster on witness-like ones).
    // scalar_mode: SCALARS_CANONICAL (`into_repr` done by the caller), SCALARS_MONT (arkworks Montgomery words, converted
    // on the device), SCALARS_WORK (the witness map's reduced-radix work form, 9 words per scalar: the h MSM)
    virtual int msm_launch(const BaseSet *bs, const u32 *d_scalars, size_t n, int scalar_mode, int c_override,
                           MsmWorkspace *ws, u32 batch = 1, size_t scalar_stride_words = 0, bool sparse = false) = 0;
    // Waits for the stream, folds the staged partial points on the host. out = `batch` XYZZ host points.
    // already_synced: the caller has synchronised with the work itself (hipGraph replay of a whole proof)
    virtual int msm_finish(MsmWorkspace *ws, HostPoint *out, bool already_synced = false) = 0;

    // The host fold of msm_finish done on the DEVICE instead (bases with precomputed multiples only: one bucket window, no
    // Horner doublings): enqueues one small kernel behind the MSM on its stream and leaves ws->batch arkworks-format XYZZ
    // points, out_stride_words u32 apart, at d_out (device memory) -- for consumers that live on the device, i.e. the
    // partial-point exchange of the sharded paths (RCCL all_gather straight from HBM, no host bounce). The workspace is
    // still handed back through msm_finish / msm_discard.
    virtual int msm_fold_device(MsmWorkspace *ws, u32 *d_out, size_t out_stride_words, hipStream_t on = nullptr) = 0;
    // an MSM whose result was taken on the device: wait for its stream, clear `pending`
    virtual int msm_discard(MsmWorkspace *ws) = 0;

    // host-point helpers (type-erased)
    virtual void hp_set_inf(HostPoint *p) const = 0;
    virtual void hp_from_affine(HostPoint *p, const u32 *affine_words) const = 0;
    virtual void hp_from_xyzz(HostPoint *p, const u32 *xyzz_words) const = 0; // arkworks-format X | Y | ZZ | ZZZ (ZZ = 0: infinity)
    virtual void hp_add(HostPoint *acc, const HostPoint *o) const = 0;
    virtual void hp_neg(HostPoint *p) const = 0;
    virtual void hp_mul(HostPoint *p, const u64 *k4) const = 0; // p = [k]p, k canonical 4x u64
    // out = [k1]p + [k2]q (shared doubling chain)
    virtual void hp_mul2(const HostPoint *p, const u64 *k1, const HostPoint *q, const u64 *k2, HostPoint *out) const = 0;
    // fixed-base table for a point that is multiplied in every proof (opaque; freed with hp_table_free)
    virtual void *hp_table_create(const HostPoint *base) const = 0;
    virtual void hp_table_mul(const void *table, const u64 *k4, HostPoint *out) const = 0;
    virtual void hp_table_free(void *table) const = 0;
    virtual void hp_to_affine(const HostPoint *p, u32 *affine_words) const = 0;
    virtual void hp_serialize(const HostPoint *p, unsigned char *out, bool compressed) const = 0;
    virtual int point_bytes(bool compressed) const = 0;

    // [k_i] * base for a batch of canonical scalars (device), result affine (device). Used for
    // synthetic base generation and key generation (fixed-base batch multiplication).
    virtual int fixed_base_mul(const u32 *base_affine_host, const u32 *d_scalars, size_t n, u32 *d_out_affine,
                               hipStream_t s) = 0;
    // element-wise group operations on host arrays of affine points (the primitive menu of
    // manta-benchmark/src/ecc.rs; op codes in mantagpu.h), run with the MSM kernels' device functions
    virtual int ec_elementwise(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_affine_host) = 0;
    // the same, results left as XYZZ points (xyzz_words() words each; no inversion on the device), and the host conversion of
    // such an array to affine points with one inversion for all of them
    virtual int ec_elementwise_xyzz(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_xyzz_host) = 0;
    virtual void xyzz_batch_to_affine(const u32 *xyzz_host, size_t n, u32 *out_affine_host) const = 0;
    // k_i P_i around other work: begin() uploads into ws's scratch buffer and launches on ws's stream, finish() waits and fetches
    // the n XYZZ results (nothing else may use ws in between)
    // glv_beta_std (G1 only; the field element beta of phi(x, y) = (beta x, y) = lambda (x, y), arkworks-format words): the
    // multipliers are then k1 + lambda k2 with k1, k2 the low two u64 of each 4 x u64 scalar -- one chain of 64 doublings
    virtual int ec_mul_xyzz_begin(const u32 *a_affine_host, const u32 *k_canonical_host, size_t n, MsmWorkspace *ws,
                                  const u32 *glv_beta_std = nullptr) = 0;
    virtual int ec_mul_xyzz_finish(MsmWorkspace *ws, size_t n, u32 *out_xyzz_host, bool glv = false) = 0;
    virtual const u32 *ec_mul_xyzz_device(MsmWorkspace *ws, size_t n, bool glv = false) const = 0; // the results in device memory (valid on ws's stream after begin())
    // radix-2 (I)NTT over a vector of 2^lg group elements (host affine in/out, natural order); d_twiddles_mont = the Fr
    // domain's omega^k table on the device, n_inv_canonical != nullptr scales by n^-1 (inverse transform)
    virtual int group_ntt(const u32 *in_affine_host, unsigned lg, const u32 *d_twiddles_mont, const u32 *n_inv_canonical,
                          u32 *out_affine_host) = 0;
    // sum of affine points (device) -> host point
    virtual int sum_affine(const u32 *d_pts, size_t n, HostPoint *out) = 0;

    MsmWorkspace *ws_acquire();
    void ws_release(MsmWorkspace *);

    static constexpr size_t MAX_IDLE_WS = 24;

  protected:
    std::mutex ws_mu_;
    std::vector<MsmWorkspace *> ws_free_;
};

GroupEngine *make_engine_bn254_g1();
GroupEngine *make_engine_bn254_g2();
GroupEngine *make_engine_bls381_g1();
GroupEngine *make_engine_bls381_g2();
GroupEngine *get_engine(int curve, int group); // cached singleton per (curve, group)

// radix sort of (key,val) pairs, keys < 2^end_bit (sort.hip: hand-written wave64 LSD radix sort)
size_t sort_pairs_temp_bytes(size_t n);
// d_count (optional): the number of pairs actually present, on the device (n is then the capacity)
bool sort_pairs_takes_device_count(int end_bit);
int sort_pairs(const u32 *keys_in, u32 *keys_out, const u32 *vals_in, u32 *vals_out, size_t n, int end_bit,
               void *tmp, size_t tmp_bytes, hipStream_t s, const u32 *d_count = nullptr, u32 lowmask = 0xffffffffu,
               u32 inv_from = 0xffffffffu); // (lowmask, inv_from): sort by key & lowmask, keys >= inv_from last -- sort.hip sort_key

} // namespace mg
