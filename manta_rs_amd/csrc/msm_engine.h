// Host orchestration of the MSM pipeline: GroupEngineT (base sets, plan, launch, finish) -- one instantiation per (curve, group).
// Part of msm_impl.h.
#pragma once
#include "msm_common.h"
#include "msm_digits.h"
#include "msm_accumulate.h"
#include "msm_reduce.h"
#include "msm_tables.h"

namespace mg {

// --------------------------------------------------------------------------------------------
// host orchestration
// --------------------------------------------------------------------------------------------
template <class Curve, int GROUP> struct GT;
template <class Curve> struct GT<Curve, 1> {
#ifdef MG_G1_SATURATED
    typedef Fp<typename Curve::Fq> F; // 32-bit saturated limbs everywhere (A/B reference build)
#else
    typedef FpR<typename Curve::Fq> F; // internal: reduced radix, lazily reduced
#endif
    typedef Fp<typename Curve::Fq> FIO; // arkworks memory format at the ABI
    typedef host::HFp<typename Curve::Fq> HF;
};
template <class Curve> struct GT<Curve, 2> {
    // G2 on the lazily-reduced Fp2R as well. Over BLS12-381 (an XYZZ point is 112 words) the 14-limb base
    // products inside Fp2R are calls (fpr_dev.h `CALLS`): fully inlined, those kernels need 256 VGPRs + 1.4 KB of
    // scratch per lane and -- observed on MI355X, ROCm 7.2 -- do not terminate. MG_G2_SATURATED keeps the
    // canonical 32-bit Fp2 path for A/B.
#ifdef MG_G2_SATURATED
    typedef Fp2<typename Curve::Fq> F;
#else
    typedef Fp2R<typename Curve::Fq> F;
#endif
    typedef Fp2<typename Curve::Fq> FIO;
    typedef host::HFp2<typename Curve::Fq> HF;
};

static inline u32 cdiv(size_t a, size_t b) { return (u32)((a + b - 1) / b); }

// The zero-fills of an MSM launch (pair counter, bucket array or direct result, timing words) as ONE kernel of ours instead of
// hipMemsetAsync calls: inside a stream capture those become memset nodes, and a memset node of a LINEAR captured graph was found
// to replay with a wrong fill pattern once other work had gone through the runtime (round 5: profiles/r05_linear_graph_defect.txt;
// the runtime pre-builds the AQL packets of such graphs, its own fill kernel included). No node of the library's graphs is a
// runtime-generated fill any more; one launch instead of two or three also shortens the chain.
// two word ranges device -> pinned host memory, a system-scope fence, then the token (msm_launch, MsmWorkspace::notify)
static __global__ __launch_bounds__(256) void stage_and_notify_kernel(const u32 *__restrict__ src0, u32 *__restrict__ dst0, u32 n0,
                                                                      const u32 *__restrict__ src1, u32 *__restrict__ dst1, u32 n1,
                                                                      u32 *__restrict__ flag) {
    for (u32 i = threadIdx.x; i < n0; i += 256) dst0[i] = src0[i];
    for (u32 i = threadIdx.x; i < n1; i += 256) dst1[i] = src1[i];
    __threadfence_system(); // every lane's stores are visible system-wide before it reaches the barrier ...
    __syncthreads();
    if (threadIdx.x == 0 && flag) {
        __atomic_store_n(flag, 1u, __ATOMIC_RELEASE); // ... and the token goes last
        __threadfence_system();
    }
}
struct ZeroRanges {
    u32 *p[3];
    u32 n[3]; // words
};
template <class F> __global__ __launch_bounds__(256) void zero_ranges(ZeroRanges r) {
    const u32 stride = gridDim.x * 256u, i0 = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        for (u32 i = i0; i < r.n[t]; i += stride) r.p[t][i] = 0u;
}

template <class Curve, int CURVE_ID, int GROUP> class GroupEngineT : public GroupEngine {
  public:
    typedef typename GT<Curve, GROUP>::F F;
    typedef typename GT<Curve, GROUP>::FIO FIO;
    typedef typename GT<Curve, GROUP>::HF HF;
    typedef host::HPoint<HF> HP;
    typedef typename Curve::Fr FrC;
    static constexpr int AW = Affine<F>::WORDS, XW = XYZZ<F>::WORDS;           // internal formats
    static constexpr int AW_IO = Affine<FIO>::WORDS, XW_IO = XYZZ<FIO>::WORDS; // arkworks formats (ABI, staging)
    static constexpr bool SAME = std::is_same<F, FIO>::value;
    // stride of one point in a BaseSet: the internal affine record padded to a multiple of 32 B (BLS12-381
    // G1: 28 -> 32 words = one 128 B line per gathered point instead of a record straddling two)
    static constexpr int AWS = SAME ? AW : (AW + 7) / 8 * 8;
    static_assert(sizeof(HP) <= sizeof(HostPoint), "HostPoint too small");

    int curve() const override { return CURVE_ID; }
    int group() const override { return GROUP; }
    int affine_words() const override { return AW_IO; }
    int xyzz_words() const override { return XW_IO; }
    int scalar_bits() const override { return FrC::BITS; }
    int base_record_bytes() const override { return AWS * 4; }
    int point_bytes(bool compressed) const override { return compressed ? HF::BYTES : 2 * HF::BYTES; }

    static HP &hp(HostPoint *p) { return *reinterpret_cast<HP *>(p); }
    static const HP &hp(const HostPoint *p) { return *reinterpret_cast<const HP *>(p); }
    void hp_set_inf(HostPoint *p) const override { hp(p) = HP::inf(); }
    void hp_from_affine(HostPoint *p, const u32 *w) const override { hp(p) = HP::from_affine_words(w); }
    void hp_from_xyzz(HostPoint *p, const u32 *w) const override { hp(p) = HP::from_xyzz_words(w); }
    void hp_add(HostPoint *a, const HostPoint *o) const override { hp(a) = HP::add(hp(a), hp(o)); }
    void hp_neg(HostPoint *p) const override { hp(p) = hp(p).neg(); }
    void hp_mul(HostPoint *p, const u64 *k4) const override { hp(p) = HP::mul(hp(p), k4, 4); }
    void hp_mul2(const HostPoint *p, const u64 *k1, const HostPoint *q, const u64 *k2, HostPoint *out) const override {
        hp(out) = HP::mul2(hp(p), k1, hp(q), k2, 4);
    }
    void *hp_table_create(const HostPoint *base) const override {
        auto *t = new host::FixedBaseTable<HP>();
        t->build(hp(base));
        return t;
    }
    void hp_table_mul(const void *table, const u64 *k4, HostPoint *out) const override {
        hp(out) = static_cast<const host::FixedBaseTable<HP> *>(table)->mul(k4);
    }
    void hp_table_free(void *table) const override { delete static_cast<host::FixedBaseTable<HP> *>(table); }
    void hp_to_affine(const HostPoint *p, u32 *w) const override { hp(p).to_affine_words(w); }
    void hp_serialize(const HostPoint *p, unsigned char *out, bool compressed) const override {
        hp(p).serialize(out, compressed);
    }

    // ---------------------------------------------------------------- bases
    int bases_create(const u32 *pts_in, size_t n_in, bool src_on_device, int pre_c, BaseSet **out,
                     bool drop_infinity = false, u32 n_sets = 1) override {
        if (!pts_in || !n_in || !out || n_sets == 0 || n_in % n_sets) return MG_ERR_ARG;
        const u32 *pts = pts_in;
        size_t n = n_in;
        std::vector<u32> compact, map;
        if (drop_infinity && !src_on_device) {
            size_t kept = 0;
            for (size_t i = 0; i < n_in; ++i) {
                const u32 *q = pts_in + i * AW_IO;
                u32 x = 0;
                for (int k = 0; k < AW_IO; ++k) x |= q[k];
                kept += x != 0;
            }
            if (kept < n_in) {
                if (kept == 0) kept = 1; // keep one infinity entry so that the set is never empty
                compact.resize(kept * AW_IO, 0u);
                map.resize(kept, 0u);
                size_t o = 0;
                for (size_t i = 0; i < n_in && o < kept; ++i) {
                    const u32 *q = pts_in + i * AW_IO;
                    u32 x = 0;
                    for (int k = 0; k < AW_IO; ++k) x |= q[k];
                    if (x != 0) {
                        std::memcpy(&compact[o * AW_IO], q, AW_IO * 4);
                        map[o++] = (u32)i;
                    }
                }
                pts = compact.data();
                n = kept;
            }
        }
        prime_occupancy();
        BaseSet *bs = new BaseSet();
        bs->curve = CURVE_ID;
        bs->group = GROUP;
        bs->device = current_device();
        bs->n = n;
        bs->n_orig = n_in;
        bs->n_sets = n_sets;
        bs->set_len = n_in / n_sets;
        if (n_sets > 1 && n_sets <= BaseSet::MAX_SETS) { // where every query starts among the stored points
            for (u32 q = 0; q <= n_sets; ++q) {
                const size_t first = (size_t)q * bs->set_len; // original index
                bs->set_first[q] = map.empty() ? (u32)(first < n ? first : n)
                                               : (u32)(std::lower_bound(map.begin(), map.end(), (u32)first) - map.begin());
            }
            bs->set_first[n_sets] = (u32)n;
        }
        if (!map.empty()) {
            if (hipMalloc((void **)&bs->d_map, map.size() * 4) != hipSuccess ||
                memcpy_sync(bs->d_map, map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                bases_destroy(bs);
                return MG_ERR_OOM;
            }
        }
        // pre_c < 0: FULL tables of window width -pre_c -- besides 2^(c w) P every multiple m 2^(c w) P, m = 1 .. 2^(c-1), so that
        // a signed digit addresses its summand directly and the MSM is one plain sum: no buckets, no sort, no bucket reduce
        const bool full = pre_c < 0;
        if (full) pre_c = -pre_c;
        int W = 1;
        if (pre_c > 0) {
            W = (FrC::BITS + pre_c - 1) / pre_c; // digits_kernel: |k| < 2^(BITS - 1)
            bs->pre_c = pre_c;
            bs->pre_W = W;
            bs->full = full;
        }
        const u32 FB = full ? 1u << (pre_c - 1) : 1u; // table entries per (window, base)
        if (full && (pre_c < 2 || pre_c > 12 || (size_t)W * n * FB >= ((size_t)1 << 31))) {
            bases_destroy(bs);
            return MG_ERR_ARG;
        }
        bs->bytes = (size_t)W * n * FB * AWS * 4;
        hipError_t e = hipMalloc((void **)&bs->d_pts, bs->bytes);
        u32 *win_pts = nullptr; // full: the window tables are an intermediate, freed below
        if (e == hipSuccess && full) e = hipMalloc((void **)&win_pts, (size_t)W * n * AWS * 4);
        if (e != hipSuccess) {
            bases_destroy(bs);
            set_last_hip_error(e, "hipMalloc(bases)", __FILE__, __LINE__);
            return MG_ERR_OOM;
        }
        struct FreeWin {
            u32 *&p;
            ~FreeWin() {
                if (p) hipFree(p);
            }
        } free_win{win_pts};
        u32 *const dst = full ? win_pts : bs->d_pts;
        if (SAME) {
            e = memcpy_sync(dst, pts, n * AW * 4, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
        } else { // convert arkworks limbs -> internal representation on the device
            u32 *stage = nullptr;
            const u32 *src = pts;
            e = hipSuccess;
            if (!src_on_device) {
                e = hipMalloc((void **)&stage, n * AW_IO * 4);
                if (e == hipSuccess) e = memcpy_sync(stage, pts, n * AW_IO * 4, hipMemcpyHostToDevice);
                src = stage;
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL((bases_to_internal<F>), dim3(cdiv(n, 256)), dim3(256), 0, setup_stream(), src, n, dst, (u32)AWS);
                e = setup_sync();
            }
            if (stage) hipFree(stage);
        }
        if (e != hipSuccess) {
            bases_destroy(bs);
            set_last_hip_error(e, "upload/convert bases", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        if (W > 1) {
            u32 *tmp = nullptr;
            const size_t cnt = (size_t)(W - 1) * n;
            e = hipMalloc((void **)&tmp, cnt * XW * 4);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "hipMalloc(precompute tmp)", __FILE__, __LINE__);
                return MG_ERR_OOM;
            }
            hipLaunchKernelGGL((precompute_chain<F>), dim3(cdiv(n, 256)), dim3(256), 0, setup_stream(), dst, (u32)AWS, (u32)n,
                               pre_c, W, tmp);
            constexpr int KB = 16;
            hipLaunchKernelGGL((xyzz_to_affine_batch<F, KB>), dim3(cdiv(cdiv(cnt, KB), 256)), dim3(256), 0, setup_stream(), tmp,
                               cnt, dst + n * AWS, (u32)AWS);
            e = setup_sync();
            hipFree(tmp);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "precompute kernels", __FILE__, __LINE__);
                return MG_ERR_HIP;
            }
        }
        if (full) { // expand the window tables, a slice of (window, base) pairs at a time (<= 512 MB of XYZZ points in flight)
            u32 *const final_pts = bs->d_pts;
            const size_t pairs = (size_t)W * n;
            size_t slice = ((size_t)512 << 20) / ((size_t)FB * XW * 4);
            if (slice < 256) slice = 256;
            if (slice > pairs) slice = pairs;
            u32 *tmp = nullptr;
            e = hipMalloc((void **)&tmp, slice * FB * XW * 4);
            constexpr int KBF = 64; // one Fermat inversion per 64 points
            for (size_t j0 = 0; e == hipSuccess && j0 < pairs; j0 += slice) {
                const size_t cntp = pairs - j0 < slice ? pairs - j0 : slice;
                hipLaunchKernelGGL((full_table_chain<F>), dim3(cdiv(cntp, 256)), dim3(256), 0, setup_stream(), win_pts, (u32)AWS, j0, (u32)cntp, FB,
                                   tmp);
                hipLaunchKernelGGL((xyzz_to_affine_batch<F, KBF>), dim3(cdiv(cdiv(cntp * FB, KBF), 256)), dim3(256), 0, setup_stream(), tmp,
                                   cntp * FB, final_pts + j0 * FB * AWS, (u32)AWS);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = setup_sync();
            else (void)setup_sync();
            if (tmp) hipFree(tmp);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "full-table kernels", __FILE__, __LINE__);
                return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
            }
        }
        *out = bs;
        return MG_OK;
    }
    void bases_destroy(BaseSet *bs) override {
        if (!bs) return;
        if (bs->d_map) hipFree(bs->d_map);
        if (bs->d_pts) hipFree(bs->d_pts);
        delete bs;
    }

    // ---------------------------------------------------------------- plan
    MsmPlan plan_for(const BaseSet *bs, size_t n, int c_override, u32 batch = 1) const override {
        MsmPlan p;
        if (bs->pre_c > 0) {
            p.c = bs->pre_c;
            p.W = bs->pre_W;
            p.precomp = true;
            p.full = bs->full;
            p.Wb = 1;
        } else {
            int lg = 0;
            while (((size_t)1 << lg) < n) ++lg;
            // (2^20 plain bases, one MSM at a time: c = 14 / 15 / 16 / 17 -> 4.98 / 4.77 / 4.48 / 5.02 ms with the front levels of the
            // bucket reduce, which 16 windows of 32 768 buckets need: profiles/r03_plain_bases_sweep.txt)
            int c = c_override > 0 ? c_override : (lg <= 8 ? 5 : lg <= 12 ? 8 : lg <= 15 ? 10 : lg <= 18 ? 12 : lg <= 19 ? 14 : 16);
            p.c = c;
            p.W = (FrC::BITS + c - 1) / c;
            p.Wb = p.W;
        }
        p.B = 1u << (p.c - 1);
        // entries per lane. Large MSMs: the grid is a whole number of rounds of 2 wavefronts per SIMD (256 CUs x 4 SIMDs x 2 x 64 =
        // 131 072 lanes) -- the accumulate kernel holds two waves per SIMD, so 1.5 rounds leave half the SIMDs idle for a third of
        // the kernel; longer chunks mean fewer partials for the merge levels, hence as few rounds as keep L <= 192. Proof-sized
        // MSMs are latency chains (L mixed additions, then the merge levels): shorter chunks, twice the lanes, never fewer than 6
        // entries each. Batched proofs: most digit entries are invalid (sorted last), so the lanes are kept plentiful (L <= 96).
        // (sweeps: profiles/history/code_comment_measurements.md "chunk length")
        const size_t M = n * (size_t)p.W * batch;
        size_t L;
        if (M < ((size_t)8 << 20)) {
            L = M / (192 * 1024);
            if (L < 6) L = 6;
            // (full tables: no sort and no bucket reduce behind the merge levels any more, and the balance moves to short chunks for
            // all five MSMs of a proof -- PrivateTransfer, sequential proof, 300 proofs per run, same box: L = 1 / 2 / 3 / 4 / 5 / 6 ->
            // 0.98-1.02 / 0.93-0.98 / 0.87-0.89 / 0.89-0.93 / 0.90-0.93 / 0.91-0.92 ms)
            if (p.full) L = 3;
        } else {
            const size_t round = 128 * 1024, lmax = batch > 1 ? 96 : 192;
            const size_t rounds = (M + round * lmax - 1) / (round * lmax);
            L = (M + round * rounds - 1) / (round * rounds);
        }
        if (const int l = ab_knob("MANTA_MSM_L", 0); l > 0) L = (size_t)l;
        p.L = (u32)L;
        return p;
    }

    // lanes of one full round of the accumulate kernel: what the device holds at the kernel's own occupancy (single MSMs: the
    // shortest chain) or at two wavefronts per SIMD (batched passes: that saturates the integer pipe, and fewer lanes mean fewer
    // partials to merge). MANTA_ACC_ROUND_WAVES = wavefronts per SIMD, 0 = off (host-side chunk length only).
    u32 acc_round_lanes(u32 batch, bool single = false) {
        static const int knob = [] {
            return ab_knob("MANTA_ACC_ROUND_WAVES", -1);
        }();
        if (knob == 0) return 0;
        const int dev = current_device();
        if (dev < 0 || dev >= 64 || !occ_[dev].cus.load(std::memory_order_acquire)) return 0; // (primed by bases_create)
        u32 w = single && occ_[dev].blocks_single ? occ_[dev].blocks_single : occ_[dev].blocks; // 256-thread blocks per CU = wavefronts per SIMD
        if (knob > 0) w = (u32)knob < w ? (u32)knob : w;
        else if (batch > 1 && w > 2) w = 2;
        return w * 256u * occ_[dev].cus.load(std::memory_order_relaxed);
    }
    struct Occ {
        u32 blocks = 0, blocks_single = 0; // accumulate_chunks / accumulate_single (more registers, LDS: its own round size)
        std::atomic<u32> cus{0};
    } occ_[64];
    // (asked once per device outside any stream capture: bases_create runs before the first MSM on its device)
    void prime_occupancy() {
        const int dev = current_device();
        if (dev < 0 || dev >= 64 || occ_[dev].cus.load(std::memory_order_acquire)) return;
        int nb = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, accumulate_chunks<F, false>, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nb < 1 || cus < 1) {
            (void)hipGetLastError();
            return;
        }
        int nbs = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_single<F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)AccSingle<F>::LDS_BYTES) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, accumulate_single<F>, 256, AccSingle<F>::LDS_BYTES) != hipSuccess || nbs < 1) {
            (void)hipGetLastError();
            nbs = 0;
        }
        std::lock_guard<std::mutex> g(side_mu_);
        occ_[dev].blocks_single = (u32)nbs;
        occ_[dev].blocks = (u32)nb;
        occ_[dev].cus.store((u32)cus, std::memory_order_release);
    }

    // few tiles = a pure latency chain: spread each addition over the workgroup's four wavefronts
    static bool coop_tiles(u32 tiles) {
        static const int lim = [] {
            return ab_knob("MANTA_COOP_TILES", 64);
        }();
        return (int)tiles <= lim;
    }
    static u32 coop_waves() { // merge levels with at most this many 64-entry waves use the cooperative kernel
        static const u32 lim = [] {
            return (u32)ab_knob("MANTA_COOP_WAVES", 512);
        }();
        return lim;
    }
    // entries folded serially per lane in the first merge level. Large MSMs: 4 (throughput). Proof-sized MSMs: 16 --
    // the level then has few enough logical waves (<= coop_waves()) for the cooperative kernel, whose additions
    // cost a third: 15 cooperative serial steps + the scan beat 3 plain steps + the scan and shrink the next level.
    static u32 merge_g1(size_t M) {
        static const u32 g = [] {
            const int v = ab_knob("MANTA_MERGE_G", 0);
            return (u32)(v >= 1 && v <= 64 ? v : 0);
        }();
        if (g) return g;
        return M < ((size_t)8 << 20) ? 16u : 4u;
    }

    // front levels of the bucket reduce (serial_reduce): 2^lgS0 items per lane while a level has >= 2^18 items, 2^lgS below
    // (MANTA_RED_S0 / MANTA_RED_S; MANTA_RED_S=0: scan kernels only; unset = 3), applied while a window segment has at
    // least min_items items (MANTA_RED_MIN); 2^lgSP items per lane in the plain sums of the Sx arrays (MANTA_RED_SP), which
    // run on a side stream next to the weighted chain unless MANTA_RED_SIDE=0.
    // History (profiles/r03_window_and_tail_study.txt): the first versions -- serial chains for the plain sums, a side stream per
    // workspace -- lost 6-9 % of the pipelined rate and were off by default; c = 20 tables (accumulate kernel 19 % shorter) still do
    // not pay: the 2^19-bucket reduce is eight more dependent launches and a third sort pass.
    struct RedKnobs {
        int lgS0, lgS, lgSP;
        u32 min_items;       // stand-alone MSMs and single proofs: a window segment takes front levels from this many buckets on
        u32 min_items_batch; // passes of several scalar vectors (batched proofs): from this many on
        bool side;
    };
    static const RedKnobs &red_knobs() {
        static const RedKnobs k = [] {
            RedKnobs r{2, -1, 3, 16384u, 2048u, true}; // lgS = -1: automatic (below)
            auto env = [](const char *n, int lo, int hi, int dflt) {
                const int v = ab_knob(n, dflt);
                return v < lo ? lo : (v > hi ? hi : v);
            };
            r.lgS0 = env("MANTA_RED_S0", 1, 8, r.lgS0);
            r.lgS = env("MANTA_RED_S", -1, 8, r.lgS);
            r.lgSP = env("MANTA_RED_SP", 1, 8, r.lgSP);
            // Batched passes (round 6, with the front levels legal inside a slot's graphs): 12-bit windows for a / b_g1 / b_g2 / l
            // (2 048 buckets per proof and MSM) and front levels from 2 048 buckets on -- the h MSM's 8 192 too -- against 11-bit
            // windows and scan tiles only: +3.4 % (W) / +4.2 % (dense) proofs/s, profiles/r06_batched_windows_front_levels.txt.
            // (an explicit MANTA_RED_MIN rules both thresholds unless MANTA_RED_MIN_BATCH says otherwise)
            const bool explicit_min = ab_knob("MANTA_RED_MIN", -1) >= 0;
            r.min_items = (u32)env("MANTA_RED_MIN", 128, 1 << 30, (int)r.min_items);
            r.min_items_batch = (u32)env("MANTA_RED_MIN_BATCH", 128, 1 << 30, explicit_min ? (int)r.min_items : (int)r.min_items_batch);
            r.side = env("MANTA_RED_SIDE", 0, 1, 1) != 0;
            return r;
        }();
        return k;
    }

    hipStream_t engine_side_stream() {
        std::lock_guard<std::mutex> g(side_mu_);
        if (!side_stream_) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                hipStreamCreateWithPriority(&side_stream_, hipStreamNonBlocking, hi) != hipSuccess)
                side_stream_ = nullptr;
        }
        return side_stream_;
    }
    std::mutex side_mu_;
    hipStream_t side_stream_ = nullptr; // process lifetime

    // ---------------------------------------------------------------- launch
    int msm_launch(const BaseSet *bs, const u32 *d_scalars, size_t n, int scalar_mode, int c_override,
                   MsmWorkspace *ws, u32 batch = 1, size_t scalar_stride_words = 0, bool sparse = false) override {
        if (!bs || !d_scalars || !ws || n == 0 || n > bs->n_orig || batch == 0 || batch > 65535) return MG_ERR_ARG;
        if (bs->curve != CURVE_ID || bs->group != GROUP) return MG_ERR_ARG;
        const bool batched_pass = batch > 1; // several scalar vectors against the same bases (a pass of several proofs)
        const u32 nsets = bs->n_sets; // concatenated queries over one scalar vector: nsets results per vector
        if (nsets > 1 && n > bs->set_len) return MG_ERR_ARG;
        const size_t n_scalars = n;        // scalars supplied by the caller (indexed by original position)
        if (bs->d_map || n > bs->n || nsets > 1) n = bs->n; // entries = stored points; the kernel zips to the shorter side
        const MsmPlan pl = plan_for(bs, n, c_override, batch);
        hipStream_t s = msm_stream_of(ws);
        const size_t M = n * (size_t)pl.W * batch;
        // full tables: a digit addresses its summand, every pair of a scalar vector carries the same key and the "bucket" is the result
        const u32 KB = pl.full ? 1u : pl.B; // bucket keys per bucket window
        if (M >= (1ull << 31) || (size_t)batch * nsets * pl.Wb * KB >= (1ull << 24)) return MG_ERR_ARG;
        if (pl.full) sparse = true; // compacting digit kernel: no invalid keys, so a single MSM needs no sort at all
        const u32 seg_keys = (u32)pl.Wb * KB; // bucket keys per (scalar vector, query)
        const u32 nb = batch * nsets * seg_keys; // real buckets; key nb = INVALID
        const u32 invalid = nb;
        int rc;
        if ((rc = ws->keys_in.reserve(M * 4)) || (rc = ws->keys_out.reserve(M * 4)) ||
            (rc = ws->vals_in.reserve(M * 4)) || (rc = ws->vals_out.reserve(M * 4)))
            return rc;
        const size_t tmpb = sort_pairs_temp_bytes(M);
        if ((rc = ws->sort_tmp.reserve(tmpb))) return rc;
        if ((rc = ws->buckets.reserve((size_t)(nb + 1) * XW * 4))) return rc;
        const u32 T = cdiv(M, pl.L);
        if ((rc = ws->pkeys[0].reserve((size_t)2 * T * 4)) || (rc = ws->ppts[0].reserve((size_t)2 * T * XW * 4)))
            return rc;
        const u32 waves1 = cdiv((size_t)2 * T, 64);
        if ((rc = ws->pkeys[1].reserve((size_t)2 * waves1 * 4)) ||
            (rc = ws->ppts[1].reserve((size_t)2 * waves1 * XW * 4)))
            return rc;

        // with precomputed tables the base index is w*stride + i: table w starts bs->n points after w-1
        if ((size_t)pl.W * bs->n * (pl.full ? pl.B : 1u) >= (1ull << 31)) return MG_ERR_ARG;
        int end_bit = 1;
        while ((1u << end_bit) <= invalid) ++end_bit;
        // the fixed layout marks a zero digit with the key `invalid` = one past the last bucket; where that key alone would cost
        // the sort another 8-bit pass (2^16 buckets: c = 17 tables) the compacting digit kernel is used instead -- its second walk
        // over the digits is a fifth of a radix pass
        int end_bit_real = 1;
        while (nb > 1 && (1u << end_bit_real) <= nb - 1) ++end_bit_real;
        if ((end_bit + 7) / 8 > (end_bit_real + 7) / 8) sparse = true;
        if (sparse) end_bit = end_bit_real; // no pair carries the invalid key there
        // Several scalar vectors in the fixed layout (the dense h MSM of a batched pass): the digit kernel writes vector q's pairs
        // behind vector q - 1's, and key = q * seg_keys + bucket with seg_keys a power of two -- a stable sort by the BUCKET bits
        // (+ one value for the invalid key) keeps every (q, bucket) run contiguous and needs bits(seg_keys) + 1 bits instead of
        // bits(batch * seg_keys) + 1: 14 instead of 19 for 32 proofs at c = 14, two radix passes over 40 M pairs instead of three
        // (sort.hip sort_key). MANTA_SORT_LOW=0: the full key (A/B).
        u32 sort_mask = 0xffffffffu, sort_inv = 0xffffffffu;
        static const bool sort_low = [] {
            return ab_knob("MANTA_SORT_LOW", 1) != 0;
        }();
        if (sort_low && !sparse && batch > 1 && nsets == 1 && (seg_keys & (seg_keys - 1)) == 0) {
            int eb = 1;
            while ((1u << eb) <= seg_keys) ++eb; // keys 0 .. seg_keys - 1, and seg_keys for the invalid ones
            if ((eb + 7) / 8 < (end_bit + 7) / 8) sort_mask = seg_keys - 1, sort_inv = invalid, end_bit = eb;
        }
        // zero digits are compacted away by the digit kernel; how many pairs remain is known on the device only
        u32 *d_count = nullptr;
        if (sparse && sort_pairs_takes_device_count(end_bit)) {
            if ((rc = ws->count.reserve(256))) return rc;
            d_count = ws->count.as<u32>();
        }
        // one key in all (full tables, one scalar vector): the run the last merge level closes IS the result -- it is stored in the
        // host's format straight away (no bucket array, no reduce launch: one node fewer on the latency chain of a proof's MSM)
#ifdef MG_NO_DIRECT // A/B builds (tools/build_variant.sh)
        const bool direct = false;
#else
        const bool direct = nb == 1;
#endif
        constexpr int XWM0 = XW > XW_IO ? XW : XW_IO;
        if (direct && ((rc = ws->redA.reserve((size_t)XWM0 * 4)) || (rc = ws->redS.reserve((size_t)XWM0 * 4)))) return rc;
        ws->timed = kernel_timing() && !ws->capturing;
        if (ws->timed && !ws->h_clk) MG_HIP(hipHostMalloc((void **)&ws->h_clk, 64, hipHostMallocDefault));
        { // every zero-fill of this launch, up front (none of the targets is touched by the digit kernel or the sort)
            ZeroRanges zr{};
            zr.p[0] = d_count, zr.n[0] = d_count ? 1u : 0u;
            // direct: no pair at all means the sum is the point at infinity; else the buckets (+ the slot of the invalid key)
            zr.p[1] = direct ? ws->redS.as<u32>() : ws->buckets.as<u32>();
            zr.n[1] = direct ? (u32)XWM0 : (u32)((size_t)(nb + 1) * XW);
            zr.p[2] = ws->timed ? (u32 *)ws->h_clk : nullptr, zr.n[2] = ws->timed ? 4u : 0u;
            const u32 most = zr.n[1] > 4u ? zr.n[1] : 4u;
            hipLaunchKernelGGL((zero_ranges<F>), dim3(most > 256u * 1024u ? 1024u : cdiv(most, 256)), dim3(256), 0, s, zr);
        }
        // Compacted pairs (witness MSMs: two thirds of the digits are zero): the host sized T for all n W digits, so the pairs
        // that remain fill an arbitrary part of it -- 1.35 rounds of wavefronts for the G2 MSM of a PrivateTransfer proof, i.e. two
        // rounds of 6 dependent additions where one round of 9 does, and 1.4 wavefronts per SIMD for a batched pass where two
        // balanced ones do. Launch one round of lanes and let the kernel derive the chunk length from the pair count.
        u32 Tl = T, adapt = 0;
        u32 Lk = pl.L; // the chunk length the kernel starts from
        // single-key MSMs sum inside the workgroup: one partial per workgroup (MANTA_ACC_SINGLE=0: the general kernel, A/B)
        // MANTA_ACC_SINGLE: bit 0 = G1, bit 1 = G2. Default G1 only (sequential PrivateTransfer proofs, sparse / W / dense, two
        // alternations on one box: off 0.770 / 0.859 / 1.258 ms, G1 0.755 / 0.852 / 1.270, G2 0.749 / 0.853 / 1.286, both 0.739 /
        // 0.863 / 1.314 -- over Fp2 the cooperative additions are ~20 us each and the dense G2 chain gets longer)
        static const bool acc_single_on = [] {
            const int v = ab_knob("MANTA_ACC_SINGLE", 1);
            return ((v >> (GROUP - 1)) & 1) != 0;
        }();
        const int dev_now = current_device();
        const bool acc_single = nb == 1 && d_count && acc_single_on && !(kernel_timing() && !ws->capturing) && dev_now >= 0 && dev_now < 64 &&
                                occ_[dev_now].cus.load(std::memory_order_acquire) && occ_[dev_now].blocks_single;
        if (d_count) {
            const u32 tgt = acc_round_lanes(batch, acc_single);
            if (tgt && Tl > tgt) Tl = tgt, adapt = 1;
            // one LARGE scalar vector (host chunk length above 6: 2^20 scalars): whatever the lane count came to, the pair count
            // decides (a batched pass that fits one round keeps its host-side chunk length: measured, -12 % otherwise)
            else if (tgt && batch == 1 && pl.L > 6) adapt = 1;
            // The kernel takes max(Lk, ceil(pairs / lanes)). The host's L is sized for ALL n W digits (2^20 scalars: 120 entries per
            // lane): on a witness of which a tenth survives the compaction it left nine SIMDs in ten idle and the others walking 120
            // dependent additions -- the 2^20 BLS12-381 G2 accumulate of BASELINE configs[2] took 7.6 ms for 0.9 M pairs
            // (profiles/r04_config2_timeline.txt). With the round of lanes fixed the pair count alone decides the chunk length.
            if (adapt && Lk > 6) Lk = 6;
        }
        static const u32 dthreads_sparse = [] {
            const int v = ab_knob("MANTA_DIGITS_THREADS", 0);
            return (u32)(v == 256 || v == 512 || v == 1024 ? v : 256); // measured: 256 beats 512 and 1024 on the same box
        }();
        const u32 dthreads = d_count ? dthreads_sparse : 256u; // compacting path: fewer, larger workgroups = fewer atomics on the counter
        // Concatenated queries on full tables, ONE scalar vector (the a | b_g1 | l MSM of a single proof): every pair's key is its
        // query. One digit launch per query, in stream order, appends query 0's pairs, then query 1's, ... -- the pairs ARE sorted
        // and the radix pass over them (histogram, two scans, scatter: 135-150 us on the chain that ends a W or dense proof) is
        // not run. MANTA_Z3_SORT=1 restores the single launch + sort (A/B).
        static const bool z3_sort = [] {
            return ab_knob("MANTA_Z3_SORT", 0) != 0;
        }();
        const bool per_query = pl.full && nsets > 1 && nsets <= BaseSet::MAX_SETS && batch == 1 && d_count && !z3_sort &&
                               bs->set_first[nsets] == (u32)bs->n;
        if (per_query) {
            for (u32 q = 0; q < nsets; ++q) {
                const u32 lo = bs->set_first[q], hi = bs->set_first[q + 1];
                if (hi <= lo) continue;
                hipLaunchKernelGGL((digits_kernel<FrC>), dim3(cdiv(hi - lo, dthreads), 1), dim3(dthreads), 0, s, d_scalars, hi, pl.c, pl.W,
                                   pl.B, 2, (u32)bs->n, scalar_mode, invalid, ws->keys_in.as<u32>(), ws->vals_in.as<u32>(),
                                   (const u32 *)bs->d_map, (u32)n_scalars, scalar_stride_words, seg_keys, d_count, nsets,
                                   (u32)bs->set_len, lo);
            }
        } else
        hipLaunchKernelGGL((digits_kernel<FrC>), dim3(cdiv(n, dthreads), batch), dim3(dthreads), 0, s, d_scalars, (u32)n, pl.c, pl.W,
                           pl.B, pl.full ? 2 : (pl.precomp ? 1 : 0), (u32)bs->n, scalar_mode, invalid,
                           ws->keys_in.as<u32>(), ws->vals_in.as<u32>(), (const u32 *)bs->d_map, (u32)n_scalars,
                           scalar_stride_words, seg_keys, d_count, nsets, (u32)bs->set_len);
        batch *= nsets; // from here on every (vector, query) pair is a vector of its own: its keys, its window sums, its result
        // one key in all (a single MSM on full tables, pairs compacted): any order is sorted; one digit launch per query: sorted
        const bool no_sort = (nb == 1 || per_query) && d_count;
        const u32 *skeys = no_sort ? ws->keys_in.as<u32>() : ws->keys_out.as<u32>();
        const u32 *svals = no_sort ? ws->vals_in.as<u32>() : ws->vals_out.as<u32>();
        if (!no_sort && (rc = sort_pairs(ws->keys_in.as<u32>(), ws->keys_out.as<u32>(), ws->vals_in.as<u32>(),
                                         ws->vals_out.as<u32>(), M, end_bit, ws->sort_tmp.p, tmpb, s, d_count, sort_mask, sort_inv)))
            return rc;
        u32 *const std_final = direct ? ws->redS.as<u32>() : (u32 *)nullptr;
        if (ws->timed) MG_HIP(hipEventRecord(ws->t0, s));
#ifdef MG_CALIBRATION
        static const bool gather_only = std::getenv("MANTA_ACC_GATHER_ONLY") != nullptr; // -DMG_CALIBRATION build only (wrong results)
        if (gather_only)
            hipLaunchKernelGGL((gather_only_chunks<F>), dim3(cdiv(T, 256)), dim3(256), 0, s, ws->keys_out.as<u32>(),
                               ws->vals_out.as<u32>(), (u32)M, pl.L, invalid, bs->d_pts, (u32)AWS, ws->pkeys[0].as<u32>(), T,
                               (const u32 *)d_count);
        else
#endif
        if (acc_single)
            hipLaunchKernelGGL((accumulate_single<F>), dim3(cdiv(Tl, 256)), dim3(256), AccSingle<F>::LDS_BYTES, s, svals, (u32)M, Lk, bs->d_pts, (u32)AWS,
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, adapt, invalid);
        else
        if (ws->timed)
            hipLaunchKernelGGL((accumulate_chunks<F, true>), dim3(cdiv(Tl, 256)), dim3(256), 0, s, skeys,
                               svals, (u32)M, Lk, invalid, bs->d_pts, (u32)AWS, ws->buckets.as<u32>(),
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, ws->h_clk, adapt);
        else
            hipLaunchKernelGGL((accumulate_chunks<F, false>), dim3(cdiv(Tl, 256)), dim3(256), 0, s, skeys,
                               svals, (u32)M, Lk, invalid, bs->d_pts, (u32)AWS, ws->buckets.as<u32>(),
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, (unsigned long long *)nullptr, adapt);
        if (ws->timed) MG_HIP(hipEventRecord(ws->t1, s));
        u32 cnt = acc_single ? cdiv(Tl, 256) : 2 * Tl;
        int src = 0;
        for (int level = 0;; ++level) {
            // entries folded serially per lane: the first level is throughput-bound (as many entries as
            // accumulate lanes x 2), later ones are pure latency; <= 512 entries finish in one wave
            u32 G = level == 0 && !acc_single ? merge_g1(M) : 2;
            if (cnt <= 512) G = cnt <= 64 ? 1 : cdiv(cnt, 64);
            const u32 waves = cdiv(cdiv(cnt, G), 64);
            const int fin = waves == 1;
            if (waves <= coop_waves())
                hipLaunchKernelGGL((merge_partials_coop<F>), dim3(waves), dim3(256), 0, s, ws->pkeys[src].as<u32>(),
                                   ws->ppts[src].as<u32>(), cnt, G, invalid, fin, ws->buckets.as<u32>(),
                                   ws->pkeys[1 - src].as<u32>(), ws->ppts[1 - src].as<u32>(), std_final);
            else
                hipLaunchKernelGGL((merge_partials<F>), dim3(cdiv(waves, 4)), dim3(256), 0, s, ws->pkeys[src].as<u32>(),
                                   ws->ppts[src].as<u32>(), cnt, G, invalid, fin, ws->buckets.as<u32>(),
                                   ws->pkeys[1 - src].as<u32>(), ws->ppts[1 - src].as<u32>(), waves, std_final);
            if (fin) break;
            cnt = 2 * waves;
            src ^= 1;
        }
        // ---- bucket reduce
        const u32 segs = batch * (u32)pl.Wb;
        // what the scan kernels below reduce: (array, points per segment, first item, items); the front levels replace
        // the bucket array by their A arrays
        const u32 *rin = ws->buckets.as<u32>();
        u32 rstride = KB, roff = 0, rn = KB, tail_shift = 0, n_extra = 0;
        u32 extra_shift[MsmWorkspace::MAX_EXTRA] = {};
        hipStream_t side = nullptr; // plain sums of the front levels run beside the weighted chain (stand-alone MSMs)
        {
            const RedKnobs &rk = red_knobs();
            // Work-efficient front levels: on (8 buckets per lane, 16 from 2^16 buckets on) wherever a window segment has >= min_items
            // buckets (profiles/r03_front_levels_ab.txt) -- since round 6 inside a proof slot's captured graphs too (the crash of
            // rounds 3-5 was the side stream, below): nothing at manta-pay sizes, whose windows stay below the threshold, -2.4 % on
            // the 2^20 proof of BASELINE configs[2] (profiles/r06_front_levels_in_graph.txt). The plain sums of STAND-ALONE MSMs ride
            // on ONE high-priority side stream per engine (a side stream per workspace aliased the runtime's shared hardware queues).
            const int lgS_eff = rk.lgS >= 0 ? rk.lgS : 3;
            const u32 min_items = batched_pass ? rk.min_items_batch : rk.min_items;
            if (lgS_eff > 0 && rn >= min_items) {
                // The side stream is for STAND-ALONE launches only, and never for a stream that is being captured. Round 6 root cause
                // (profiles/r06_front_levels_in_graph.txt): inside the forked capture of a proof slot the four G1 MSMs are four
                // branches, and the ONE side stream of the engine was forked from and joined into each of them in turn -- the
                // runtime's per-stream lists of "parallel capture streams" became cyclic (branch a <-> side <-> branch b) and
                // hipStreamEndCapture recursed over them until the stack was gone (SIGSEGV in hip::Stream::EndCapture, 25+ frames
                // of itself). That was the "pass fails" of profiles/r05_batched_ab.txt (3) and the reason behind in_graph_slot.
                hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
                const bool being_captured = hipStreamIsCapturing(s, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
                if (!ws->capturing && !ws->run_on && !ws->in_graph_slot && !being_captured && rk.side) {
                    // ONE side stream per engine, high priority (= the runtime's other pool of hardware queues)
                    if (!(side = engine_side_stream())) return MG_ERR_HIP;
                    if (!ws->side_fork) {
                        MG_HIP(hipEventCreateWithFlags(&ws->side_fork, hipEventDisableTiming));
                        MG_HIP(hipEventCreateWithFlags(&ws->side_join, hipEventDisableTiming));
                    }
                    ws->side_stream = side; // (for the abandon paths: they drain it; not owned by the workspace)
                }
                // one level: lanes of 2^lg items; cooperative additions when the level has few lanes
                auto level = [&](hipStream_t st, const u32 *in, u32 stride, u32 off, u32 n, int lg, u32 lanes, u32 *A, u32 *Sx) {
                    const size_t nl = (size_t)segs * lanes;
                    if (cdiv(nl, 64) <= coop_waves())
                        hipLaunchKernelGGL((serial_reduce_coop<F>), dim3(cdiv(nl, 64)), dim3(256), 0, st, in, stride, off, n, 1u << lg,
                                           lanes, (u32)nl, A, Sx);
                    else
                        hipLaunchKernelGGL((serial_reduce<F>), dim3(cdiv(nl, 256)), dim3(256), 0, st, in, stride, off, n, 1u << lg,
                                           lanes, (u32)nl, A, Sx);
                };
                // two passes over the same loop: sizes first (one reservation), then the launches
                for (int pass = 0; pass < 2; ++pass) {
                    size_t used = 0; // points
                    auto take = [&](size_t pts) {
                        u32 *p = pass ? ws->front.as<u32>() + used * XW : nullptr;
                        used += pts;
                        return p;
                    };
                    const u32 *in = ws->buckets.as<u32>();
                    u32 stride = pl.B, off = 0, n = pl.B, shift = 0, ne = 0;
                    while (n >= min_items && ne < (u32)MsmWorkspace::MAX_EXTRA) {
                        // the big first levels are throughput-bound: short stretches = enough lanes for two wavefronts per SIMD;
                        // below that a level is a latency chain either way and longer stretches save a level (2^16 buckets: 16 per lane
                        // leaves the scan kernels the 4 096 items they take at c = 16)
                        const int lg = (size_t)segs * n >= ((size_t)1 << 18) ? rk.lgS0 : (rk.lgS < 0 && n >= (1u << 16) ? 4 : lgS_eff);
                        const u32 lanes = cdiv(n, 1u << lg);
                        u32 *A = take((size_t)segs * lanes), *Sx = take((size_t)segs * lanes);
                        if (pass) level(s, in, stride, off, n, lg, lanes, A, Sx);
                        // plain sum of the Sx_t: serial partial sums until one tile per segment is left, then one wavefront
                        hipStream_t ps = side ? side : s;
                        if (pass && side) {
                            MG_HIP(hipEventRecord(ws->side_fork, s));
                            MG_HIP(hipStreamWaitEvent(side, ws->side_fork, 0));
                        }
                        const u32 *pin = Sx;
                        u32 pcnt = lanes;
                        while (pcnt > 64) {
                            int plg = rk.lgSP;
                            while (plg > 1 && (pcnt >> plg) < 32 && pcnt > 64u << 1) --plg; // do not shrink below a tile
                            const u32 pl2 = cdiv(pcnt, 1u << plg);
                            u32 *t = take((size_t)segs * pl2);
                            if (pass) level(ps, pin, pcnt, 0u, pcnt, plg, pl2, t, (u32 *)nullptr);
                            pin = t;
                            pcnt = pl2;
                        }
                        if (pass) {
                            u32 *dst = ws->extra.as<u32>() + (size_t)ne * segs * XW_IO;
                            if (coop_tiles(segs))
                                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs), dim3(256), 0, ps, pin, pcnt, 0u, pcnt, 1u, dst,
                                                   (u32 *)nullptr, 1);
                            else
                                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs, 4)), dim3(256), 0, ps, pin, pcnt, 0u, pcnt,
                                                   1u, segs, dst, (u32 *)nullptr, 1);
                        }
                        extra_shift[ne++] = shift;
                        shift += lg;
                        in = A, stride = lanes, off = 1, n = lanes - 1;
                    }
                    if (!pass) {
                        if ((rc = ws->front.reserve(used * XW * 4)) ||
                            (rc = ws->extra.reserve((size_t)MsmWorkspace::MAX_EXTRA * segs * XW_IO * 4)))
                            return rc;
                    } else {
                        rin = in, rstride = stride, roff = off, rn = n, tail_shift = shift, n_extra = ne;
                    }
                }
            }
        }
        const u32 T0 = cdiv(rn, 64);
        u32 T1 = 0, nP = 0;
        size_t stage_pts;
        constexpr int XWM = XW > XW_IO ? XW : XW_IO;
        // small results (every MSM of a proof) leave through stage_and_notify_kernel; large ones keep the runtime's copy
        auto own_stage = [&](size_t pts) { return ws->notify || (pts + (size_t)n_extra * segs) * XW_IO <= 16384; };
        if (direct) { // the last merge level left the result in redS
            stage_pts = segs;
            if ((rc = stage_reserve(ws, stage_pts * XW_IO * 4))) return rc;
            ws->d_tail = ws->redS.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->redS.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        } else if (T0 == 1) { // a single tile per window: its S is the window sum
            if ((rc = ws->redA.reserve((size_t)segs * XWM * 4)) || (rc = ws->redS.reserve((size_t)segs * XWM * 4))) return rc;
            if (coop_tiles(segs) && rn > 1) // (rn = 1, full tables: the scan kernel has no addition to make, it converts the point)
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs), dim3(256), 0, s, rin, rstride, roff, rn, 1u,
                                   ws->redA.as<u32>(), ws->redS.as<u32>(), 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs, 4)), dim3(256), 0, s, rin, rstride, roff, rn, 1u, segs, ws->redA.as<u32>(), ws->redS.as<u32>(), 1);
            stage_pts = segs;
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->redS.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->redS.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        } else if (T0 <= 64) { // two launches: tiles, then (X, sumS) per window
            if ((rc = ws->redA.reserve((size_t)segs * T0 * XW * 4)) || (rc = ws->redS.reserve((size_t)segs * T0 * XW * 4)) ||
                (rc = ws->misc.reserve((size_t)segs * 2 * XW_IO * 4)))
                return rc;
            if (coop_tiles(segs * T0))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * T0), dim3(256), 0, s, rin, rstride, roff, rn,
                                   T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T0, 4)), dim3(256), 0, s, rin, rstride, roff, rn, T0, segs * T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            if (coop_tiles(segs * 2))
                hipLaunchKernelGGL((reduce_level1_coop<F>), dim3(segs * 2), dim3(256), 0, s, ws->redA.as<u32>(), ws->redS.as<u32>(),
                                   T0, ws->misc.as<u32>());
            else
                hipLaunchKernelGGL((reduce_level1<F>), dim3(segs), dim3(128), 0, s, ws->redA.as<u32>(), ws->redS.as<u32>(), T0,
                                   ws->misc.as<u32>());
            stage_pts = (size_t)segs * 2;
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->misc.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->misc.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
            T1 = 0xffffffffu; // marks the (X, sumS) layout for msm_finish
        } else {
            if ((rc = ws->redA.reserve((size_t)segs * T0 * XWM * 4)) || (rc = ws->redS.reserve((size_t)segs * T0 * XWM * 4)))
                return rc;
            hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T0, 4)), dim3(256), 0, s, rin, rstride, roff, rn, T0, segs * T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            T1 = cdiv(T0 - 1, 64); // level 1 over A0[1..T0-1]
            nP = cdiv(T0, 64);     // plain sums of S0[0..T0-1]
            if ((rc = ws->misc.reserve((size_t)segs * (2 * T1 + nP) * XW_IO * 4))) return rc;
            u32 *A1 = ws->misc.as<u32>();
            u32 *S1 = A1 + (size_t)segs * T1 * XW_IO;
            u32 *P0 = S1 + (size_t)segs * T1 * XW_IO;
            if (coop_tiles(segs * T1))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * T1), dim3(256), 0, s, ws->redA.as<u32>(), T0, 1u, T0 - 1, T1,
                                   A1, S1, 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T1, 4)), dim3(256), 0, s,
                                   ws->redA.as<u32>(), T0, 1u, T0 - 1, T1, segs * T1, A1, S1, 1);
            if (coop_tiles(segs * nP))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * nP), dim3(256), 0, s, ws->redS.as<u32>(), T0, 0u, T0, nP, P0,
                                   (u32 *)nullptr, 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * nP, 4)), dim3(256), 0, s,
                                   ws->redS.as<u32>(), T0, 0u, T0, nP, segs * nP, P0, (u32 *)nullptr, 1);
            stage_pts = (size_t)segs * (2 * T1 + nP);
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->misc.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->misc.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        }
        if (n_extra && side) { // the plain sums ran on the side stream: join
            MG_HIP(hipEventRecord(ws->side_join, side));
            MG_HIP(hipStreamWaitEvent(s, ws->side_join, 0));
        }
        if (n_extra && !own_stage(stage_pts))
            MG_HIP(hipMemcpyAsync((u32 *)ws->h_stage + stage_pts * XW_IO, ws->extra.p, (size_t)n_extra * segs * XW_IO * 4,
                                  hipMemcpyDeviceToHost, s));
        ws->tail_shift = tail_shift;
        ws->n_extra = n_extra;
        for (u32 e = 0; e < n_extra; ++e) ws->extra_shift[e] = extra_shift[e];
        ws->extra_off_pts = stage_pts;
        if (own_stage(stage_pts)) {
            // The staged points leave through a kernel of ours (no copy node of the runtime's in a captured graph), and where the host
            // polls for the end of this chain, the same kernel raises the token. The host polls *h_flag to learn that THIS chain has ended (prover.cpp finish_pass_body). Rounds 4-5 wrote the staged
            // points with one D2H copy and the token with a second one behind it in the same stream: stream order says when each
            // copy may START, not in which order two different dispatches' writes become visible to a host that polls memory -- the
            // soak (tools/soak.py, distinct assignments) caught one single proof in ~10^5 whose a / l sum was read before it had
            // arrived (A and C, or C alone, wrong; status 0). One kernel now writes the staged points to pinned memory, fences at
            // system scope, and only then writes the token.
            if (ws->notify && !ws->h_flag) {
                MG_HIP(hipHostMalloc((void **)&ws->h_flag, 64, hipHostMallocDefault));
                *ws->h_flag = 0;
            }
            hipLaunchKernelGGL(stage_and_notify_kernel, dim3(1), dim3(256), 0, s, ws->d_tail, (u32 *)ws->h_stage, (u32)(stage_pts * XW_IO),
                               (const u32 *)ws->extra.p, (u32 *)ws->h_stage + stage_pts * XW_IO, (u32)((size_t)n_extra * segs * XW_IO),
                               ws->notify ? ws->h_flag : (u32 *)nullptr);
        }
        if (!ws->capturing) MG_HIP(hipEventRecord(ws->done, s));
        MG_HIP(hipGetLastError());
        ws->plan = pl;
        ws->T1 = T1;
        ws->nP = nP;
        ws->batch = batch;
        ws->pending = 1;
        return MG_OK;
    }

    static int stage_reserve(MsmWorkspace *ws, size_t bytes) {
        if (ws->h_stage_cap >= bytes) return MG_OK;
        if (ws->h_stage) hipHostFree(ws->h_stage);
        ws->h_stage = nullptr;
        ws->h_stage_cap = 0;
        size_t cap = bytes < 65536 ? 65536 : bytes;
        MG_HIP(hipHostMalloc(&ws->h_stage, cap, hipHostMallocDefault));
        ws->h_stage_cap = cap;
        return MG_OK;
    }

    // ---------------------------------------------------------------- finish (host fold)
    int msm_finish(MsmWorkspace *ws, HostPoint *out, bool already_synced = false) override {
        if (!ws || !ws->pending) return MG_ERR_STATE;
        if (!already_synced) MG_HIP(hipEventSynchronize(ws->done));
        ws->pending = 0;
        if (ws->timed && !already_synced) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ws->t0, ws->t1) == hipSuccess) set_last_accumulate_ms(ms);
            const unsigned long long ck[2] = {((volatile unsigned long long *)ws->h_clk)[0], ((volatile unsigned long long *)ws->h_clk)[1]};
            int khz = 0, dev = 0;
            if (ck[1] && hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess)
                set_last_accumulate_mhz((float)((double)ck[0] / (double)ck[1] * (double)khz / 1e3));
        }
        const MsmPlan &pl = ws->plan;
        const u32 Wb = (u32)pl.Wb, segs = ws->batch * Wb, T1 = ws->T1, nP = ws->nP;
        const u32 *st = (const u32 *)ws->h_stage;
        for (u32 q = 0; q < ws->batch; ++q) {
        HP total = HP::inf();
        for (int w = (int)((q + 1) * Wb) - 1; w >= (int)(q * Wb); --w) {
            HP win;
            if (T1 == 0xffffffffu) { // fused reduce: (X, sumS) per window, window = sumS + 64 X
                const HP X = HP::from_xyzz_words(st + ((size_t)w * 2 + 0) * XW_IO);
                const HP sumS = HP::from_xyzz_words(st + ((size_t)w * 2 + 1) * XW_IO);
                win = HP::add(sumS, HP::mul_pow2(X, 6));
            } else if (T1 == 0) { // one tile per window: the staged point is the window sum
                win = HP::from_xyzz_words(st + (size_t)w * XW_IO);
            } else {
                const u32 *A1 = st + ((size_t)w * T1) * XW_IO;
                const u32 *S1 = st + ((size_t)segs * T1 + (size_t)w * T1) * XW_IO;
                const u32 *P0 = st + ((size_t)segs * 2 * T1 + (size_t)w * nP) * XW_IO;
                // X = sum_{t>=1} t*A0[t] = sum_u ( S1[u] + 64*u*A1[u] )
                HP sumS = HP::inf(), run = HP::inf(), uA = HP::inf();
                for (int u = (int)T1 - 1; u >= 0; --u) {
                    sumS = HP::add(sumS, HP::from_xyzz_words(S1 + (size_t)u * XW_IO));
                    if (u >= 1) {
                        run = HP::add(run, HP::from_xyzz_words(A1 + (size_t)u * XW_IO));
                        uA = HP::add(uA, run); // sum_u u*A1[u]
                    }
                }
                HP X = HP::add(sumS, HP::mul_pow2(uA, 6));
                HP sumP = HP::inf();
                for (u32 u = 0; u < nP; ++u) sumP = HP::add(sumP, HP::from_xyzz_words(P0 + (size_t)u * XW_IO));
                win = HP::add(sumP, HP::mul_pow2(X, 6));
            }
            if (ws->tail_shift) win = HP::mul_pow2(win, ws->tail_shift); // front levels: window = 2^shift * tail + extras
            for (u32 e = 0; e < ws->n_extra; ++e) {
                const HP x = HP::from_xyzz_words(st + (ws->extra_off_pts + (size_t)e * segs + (size_t)w) * XW_IO);
                win = HP::add(win, ws->extra_shift[e] ? HP::mul_pow2(x, ws->extra_shift[e]) : x);
            }
            if (w != (int)((q + 1) * Wb) - 1) total = HP::mul_pow2(total, (unsigned)pl.c);
            total = HP::add(total, win);
        }
        hp(out + q) = total;
        }
        return MG_OK;
    }

    // ---------------------------------------------------------------- finish on the device
    int msm_fold_device(MsmWorkspace *ws, u32 *d_out, size_t out_stride_words, hipStream_t on = nullptr) override {
        if (!ws || !ws->pending || !d_out || !ws->d_tail) return MG_ERR_STATE;
        const MsmPlan &pl = ws->plan;
        if (pl.Wb != 1) return MG_ERR_STATE; // plain bases keep the host fold (up to 255 Horner doublings: a host job)
        FoldDesc d{};
        d.tail = ws->d_tail;
        d.extra = ws->extra.as<u32>();
        d.kind = ws->T1 == 0xffffffffu ? 1u : (ws->T1 == 0 ? 0u : 2u);
        d.T1 = ws->T1, d.nP = ws->nP, d.segs = ws->batch, d.n_extra = ws->n_extra, d.tail_shift = ws->tail_shift;
        for (u32 e = 0; e < ws->n_extra; ++e) d.extra_shift[e] = ws->extra_shift[e];
        hipStream_t s = on ? on : (msm_stream_of(ws));
        hipLaunchKernelGGL((fold_windows<F>), dim3(ws->batch), dim3(64), 0, s, d, d_out, out_stride_words);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int msm_discard(MsmWorkspace *ws) override {
        if (!ws) return MG_ERR_STATE;
        ws->pending = 0;
        MG_HIP(hipStreamSynchronize(msm_stream_of(ws)));
        return MG_OK;
    }

    // ---------------------------------------------------------------- fixed-base batch mul
    int fixed_base_mul(const u32 *base_affine_host, const u32 *d_scalars, size_t n, u32 *d_out_affine,
                       hipStream_t s) override {
        if (!s) s = setup_stream(); // (never the NULL stream: engine.h)
        u32 *d_base = nullptr, *tmp = nullptr;
        MG_HIP(hipMalloc((void **)&d_base, AW_IO * 4));
        hipError_t e = hipMalloc((void **)&tmp, n * XW_IO * 4);
        if (e != hipSuccess) {
            hipFree(d_base);
            set_last_hip_error(e, "hipMalloc(fixed_base tmp)", __FILE__, __LINE__);
            return MG_ERR_OOM;
        }
        hipMemcpyAsync(d_base, base_affine_host, AW_IO * 4, hipMemcpyHostToDevice, s);
        constexpr int KB = 16;
        static const size_t table_min = [] {
            return (size_t)ab_knob("MANTA_FIXED_BASE_TABLE_MIN", 16384);
        }();
        u32 *t_xyzz = nullptr, *t_aff = nullptr;
        if (n >= table_min) { // many multiples of one base: 32 table additions each instead of ~380 group operations
            constexpr size_t TN = 32 * 255;
            if (hipMalloc((void **)&t_xyzz, TN * XW_IO * 4) == hipSuccess && hipMalloc((void **)&t_aff, TN * AW_IO * 4) == hipSuccess) {
                hipLaunchKernelGGL((fixed_base_table_kernel<FIO>), dim3(cdiv(TN, 256)), dim3(256), 0, s, d_base, t_xyzz);
                hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(TN, KB), 256)), dim3(256), 0, s, t_xyzz, TN, t_aff,
                                   (u32)AW_IO);
                hipLaunchKernelGGL((fixed_base_mul_table_kernel<FIO>), dim3(cdiv(n, 256)), dim3(256), 0, s, t_aff, d_scalars, n, tmp);
            } else {
                (void)hipGetLastError();
                if (t_xyzz) hipFree(t_xyzz);
                t_xyzz = nullptr;
            }
        }
        if (!t_xyzz)
        hipLaunchKernelGGL((fixed_base_mul_kernel<FIO>), dim3(cdiv(n, 256)), dim3(256), 0, s, d_base, d_scalars, n, tmp);
        hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, s, tmp, n,
                           d_out_affine, (u32)AW_IO);
        e = hipStreamSynchronize(s);
        hipFree(d_base);
        hipFree(tmp);
        if (t_xyzz) hipFree(t_xyzz);
        if (t_aff) hipFree(t_aff);
        if (e != hipSuccess) {
            set_last_hip_error(e, "fixed_base_mul", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }

    int ec_elementwise(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_affine_host) override {
        return ec_elementwise_impl(op, a_host, b_host, n, out_affine_host, false);
    }
    // the same results as XYZZ points (XW_IO words each): no inversion on the device -- xyzz_batch_to_affine() turns them
    // into affine points on the host with one inversion for all of them
    int ec_elementwise_xyzz(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_xyzz_host) override {
        return ec_elementwise_impl(op, a_host, b_host, n, out_xyzz_host, true);
    }
    // the multiplication k_i P_i (op MG_EC_MUL) as two calls around other work: begin() uploads into the workspace's grow-only
    // scratch buffer and launches on the workspace's stream (no hipMalloc / hipFree / stream 0: nothing else on the device
    // waits for it and it waits for nothing), finish() waits and fetches the XYZZ results
    int ec_mul_xyzz_begin(const u32 *a_host, const u32 *k_host, size_t n, MsmWorkspace *ws, const u32 *glv_beta_std) override {
        if (!a_host || !k_host || !n || !ws) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv_beta_std ? (size_t)AW_IO / 2 * 4 : 0), tb = n * XW_IO * 4;
        int rc = ws->scratch.reserve(ab + bb + tb);
        if (rc) return rc;
        unsigned char *d = (unsigned char *)ws->scratch.p;
        hipError_t e = hipMemcpyAsync(d, a_host, ab, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + ab, k_host, n * 32, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess && glv_beta_std)
            e = hipMemcpyAsync(d + ab + n * 32, glv_beta_std, (size_t)AW_IO / 2 * 4, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((ec_elementwise_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, ws->stream, glv_beta_std ? 6 : 3, (const u32 *)d,
                               (const u32 *)(d + ab), n, (u32 *)(d + ab + bb));
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            hipStreamSynchronize(ws->stream);
            set_last_hip_error(e, "ec_mul_xyzz", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }
    const u32 *ec_mul_xyzz_device(MsmWorkspace *ws, size_t n, bool glv) const override { // where begin()'s kernel leaves the n results
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv ? (size_t)AW_IO / 2 * 4 : 0);
        return ws && ws->scratch.p ? (const u32 *)((unsigned char *)ws->scratch.p + ab + bb) : nullptr;
    }
    int ec_mul_xyzz_finish(MsmWorkspace *ws, size_t n, u32 *out_xyzz_host, bool glv) override {
        if (!ws || !n || !out_xyzz_host) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv ? (size_t)AW_IO / 2 * 4 : 0), tb = n * XW_IO * 4;
        hipError_t e = hipMemcpyAsync(out_xyzz_host, (unsigned char *)ws->scratch.p + ab + bb, tb, hipMemcpyDeviceToHost, ws->stream);
        const hipError_t e2 = hipStreamSynchronize(ws->stream);
        if (e == hipSuccess) e = e2;
        if (e != hipSuccess) {
            set_last_hip_error(e, "ec_mul_xyzz", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }
    void xyzz_batch_to_affine(const u32 *xyzz_host, size_t n, u32 *out_affine_host) const override {
        typedef decltype(HP{}.x) HF;
        std::vector<HF> den(n), pre(n);
        HF acc = HF::one();
        for (size_t i = 0; i < n; ++i) { // Montgomery's trick: prefix products of the denominators ZZ ZZZ (1 for infinity)
            const HP q = HP::from_xyzz_words(xyzz_host + i * XW_IO);
            den[i] = q.is_inf() ? HF::one() : HF::mul(q.zz, q.zzz);
            pre[i] = acc;
            acc = HF::mul(acc, den[i]);
        }
        HF inv = HF::inv(acc);
        for (size_t i = n; i-- > 0;) {
            const HP q = HP::from_xyzz_words(xyzz_host + i * XW_IO);
            const HF t = HF::mul(inv, pre[i]); // 1 / (ZZ ZZZ) of point i
            inv = HF::mul(inv, den[i]);
            u32 *o = out_affine_host + i * AW_IO;
            if (q.is_inf()) {
                std::memset(o, 0, AW_IO * 4);
                continue;
            }
            HF::mul(q.x, HF::mul(t, q.zzz)).store_words(o);
            HF::mul(q.y, HF::mul(t, q.zz)).store_words(o + HF::WORDS);
        }
    }
    int ec_elementwise_impl(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_host, bool xyzz) {
        if (op < 0 || op > 5 || !a_host || !out_host || n == 0 || (op != 2 && !b_host)) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = op == 3 ? n * 32 : (op == 5 ? 32 : ab);
        u32 *da = nullptr, *db = nullptr, *tmp = nullptr, *dout = nullptr;
        // a stream of its own (not stream 0: a synchronous copy anywhere else in the process -- another thread creating a base
        // set, say -- would wait for this kernel, a millisecond of one-lane latency for 128-bit multipliers)
        hipStream_t st = stream_pool_get_normal(); // (pooled: the library destroys no stream, runtime.cpp)
        hipError_t e = st ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void **)&da, ab);
        if (e == hipSuccess) e = hipMalloc((void **)&db, bb);
        if (e == hipSuccess) e = hipMalloc((void **)&tmp, n * XW_IO * 4);
        if (e == hipSuccess && !xyzz) e = hipMalloc((void **)&dout, ab);
        if (e == hipSuccess) e = hipMemcpyAsync(da, a_host, ab, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && op != 2) e = hipMemcpyAsync(db, b_host, bb, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((ec_elementwise_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, st, op, da, db, n, tmp);
            if (xyzz) {
                e = hipMemcpyAsync(out_host, tmp, n * XW_IO * 4, hipMemcpyDeviceToHost, st);
            } else {
                constexpr int KB = 16;
                hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, st, tmp, n, dout,
                                   (u32)AW_IO);
                e = hipMemcpyAsync(out_host, dout, ab, hipMemcpyDeviceToHost, st);
            }
        }
        if (st) {
            const hipError_t e2 = hipStreamSynchronize(st);
            if (e == hipSuccess) e = e2;
            stream_pool_put_normal(st);
        }
        hipFree(da);
        hipFree(db);
        hipFree(tmp);
        if (dout) hipFree(dout);
        if (e != hipSuccess) {
            set_last_hip_error(e, "ec_elementwise", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }

    // NTT over group elements: host affine in, host affine out (natural order both); tw = the Fr domain's device twiddle
    // table (omega^k, k < n/2, Montgomery), n_inv_canonical = n^-1 for the inverse transform (nullptr: forward)
    int group_ntt(const u32 *in_affine_host, unsigned lg, const u32 *d_twiddles_mont, const u32 *n_inv_canonical,
                  u32 *out_affine_host) override {
        if (!in_affine_host || !out_affine_host || lg > 26 || (lg > 0 && !d_twiddles_mont)) return MG_ERR_ARG;
        const size_t n = (size_t)1 << lg, ab = n * AW_IO * 4;
        u32 *d_in = nullptr, *d_pts = nullptr, *d_std = nullptr, *d_out = nullptr, *d_sc = nullptr;
        hipError_t e = hipMalloc((void **)&d_in, ab);
        if (e == hipSuccess) e = hipMalloc((void **)&d_pts, n * XW * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_std, n * XW_IO * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, ab);
        if (e == hipSuccess && n_inv_canonical) e = hipMalloc((void **)&d_sc, 32);
        if (e == hipSuccess) e = memcpy_sync(d_in, in_affine_host, ab, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_inv_canonical) e = memcpy_sync(d_sc, n_inv_canonical, 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((group_ntt_load_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, setup_stream(), d_in, lg, d_pts);
            for (unsigned s = 1; s <= lg; ++s)
                hipLaunchKernelGGL((group_ntt_stage_kernel<F, FrC>), dim3(cdiv(n / 2, 256)), dim3(256), 0, setup_stream(), d_pts, d_twiddles_mont, lg, s);
            hipLaunchKernelGGL((group_scale_store_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, setup_stream(), d_pts, (const u32 *)d_sc, n, d_std);
            constexpr int KB = 16;
            hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, setup_stream(), d_std, n, d_out, (u32)AW_IO);
            e = memcpy_sync(out_affine_host, d_out, ab, hipMemcpyDeviceToHost);
        }
        hipFree(d_in), hipFree(d_pts), hipFree(d_std), hipFree(d_out), hipFree(d_sc);
        if (e != hipSuccess) {
            set_last_hip_error(e, "group_ntt", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }

    int sum_affine(const u32 *d_pts, size_t n, HostPoint *out) override {
        const u32 T = n < 4096 ? (u32)(n ? n : 1) : 4096;
        u32 *tmp = nullptr;
        MG_HIP(hipMalloc((void **)&tmp, (size_t)T * XW_IO * 4));
        hipLaunchKernelGGL((sum_affine_kernel<FIO>), dim3(cdiv(T, 256)), dim3(256), 0, setup_stream(), d_pts, n, T, tmp);
        std::vector<u32> h((size_t)T * XW_IO);
        hipError_t e = memcpy_sync(h.data(), tmp, h.size() * 4, hipMemcpyDeviceToHost);
        hipFree(tmp);
        if (e != hipSuccess) {
            set_last_hip_error(e, "sum_affine", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        HP acc = HP::inf();
        for (u32 t = 0; t < T; ++t) acc = HP::add(acc, HP::from_xyzz_words(h.data() + (size_t)t * XW_IO));
        hp(out) = acc;
        return MG_OK;
    }
};

} // namespace mg
