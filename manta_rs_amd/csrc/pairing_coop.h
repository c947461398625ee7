// Wave-cooperative pairing: ONE WAVEFRONT per pairing, the Fq12 state in LDS, one Fq2 product per lane.
//
// pairing_dev.h runs a pairing on one lane; that is the right shape for hundreds of independent pairings but leaves a
// single Groth16 verification (three Miller loops and one final exponentiation) with 54 dependent Fq
// products per Fq12 product: 45.7 ms per verification on MI355X, 26.9 ms of it the final exponentiation on ONE lane
// (profiles/r02_verify_*.txt). Here an Fq12 is a polynomial of degree < 6 in w over Fq2 (w^6 = xi, w^2 = v of the
// arkworks tower), a product is the 36 coefficient products a_i b_j on 36 lanes followed by six lanes folding them
// (c_m = sum_{i+j=m} + xi sum_{i+j=m+6}), i.e. ONE Fq2 product deep instead of eighteen; line multiplications use 18 lanes,
// Frobenius maps six, the G2 doubling / addition steps of G2Prepared up to five. The arithmetic (canonical saturated
// Montgomery Fp<> of fp_dev.h) and every formula are those of pairing_dev.h, so results are the same bytes -- the
// committed verifying keys of the reference stay the known-answer test.
//
// LDS layout: slots of one Fq2 (2N words). Slots 0..35 hold the coefficient products; Fq12 register r occupies the six
// slots REG0 + 6 r .. in arkworks' memory order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (slot s = 3 i + j holds the
// coefficient of w^(i + 2 j)). A workgroup is exactly one wavefront, so __syncthreads() is a wave-level fence.
#pragma once
#include "pairing_dev.h"

namespace mg {

extern __shared__ u32 mg_pairing_lds[];

template <class K> struct PairingWave {
    typedef Pairing<K> P;
    typedef typename P::F F;
    typedef typename P::F2 F2;
    static constexpr int W = P::F2W, N = P::N;
    static constexpr int PROD = 0, REG0 = 36;
    static constexpr int R(int r) { return REG0 + 6 * r; }
    static constexpr size_t lds_bytes(int regs) { return (size_t)(REG0 + 6 * regs) * W * 4; }
    // (everything is static and addresses the workgroup's dynamic LDS block directly: the compiler then emits ds_read /
    // ds_write instead of flat accesses, and the out-of-line functions take their few integer arguments in registers)
    static MG_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
    static MG_DEV int wave_id() { return (int)(threadIdx.x >> 6); }
    // LDS block of a workgroup: [wave 0: products + its registers][wave 1 of the Miller kernel: the G2 arithmetic's slots]
    // [ring of RING coefficient triples][produced, consumed]
    static constexpr int MILLER_WORDS = (REG0 + 6) * W, RING = 4;
    static MG_DEV u32 *base() { return mg_pairing_lds + wave_id() * MILLER_WORDS; }
    static MG_DEV F2 ld(int slot) { return F2::load(base() + slot * W); }
    static MG_DEV void st(int slot, const F2 &v) { v.store(base() + slot * W); }
    // Fq2 products inlined down to the Fq product (a real function with its operands in VGPRs, fp_dev.h); pairing_dev.h's
    // out-of-line Fq2 helpers pass their operands through scratch memory, which costs more than the arithmetic here
    // (by value: operands and result travel in VGPRs; the three Karatsuba products run interleaved, Fp::mul_many)
    static __device__ __noinline__ F2 mul2(const F2 a, const F2 b) {
        const F x[3] = {a.c0, a.c1, F::add(a.c0, a.c1)}, y[3] = {b.c0, b.c1, F::add(b.c0, b.c1)};
        F v[3];
        F::template mul_many<3>(x, y, v);
        return F2{F::sub(v[0], v[1]), F::sub(F::sub(v[2], v[0]), v[1])};
    }
    static __device__ __noinline__ F2 mul2_fp(const F2 a, const F k) {
        const F x[2] = {a.c0, a.c1}, y[2] = {k, k};
        F v[2];
        F::template mul_many<2>(x, y, v);
        return F2{v[0], v[1]};
    }
    static MG_DEV F2 sqr2(const F2 &a) { return mul2(a, a); }
    // 9 a mod p as ONE small multiple: t = 9 a (N + 1 words), a quotient estimate q from its top 64 bits (never above
    // floor(t / p), at most one below: the divisor is p's top word + 1 and the float product is biased down by 2^-20), then
    // t - q p < 2p and one conditional subtraction -- ~45 instructions against 4 modular additions (~150): xi = 9 + u is
    // applied to every wrapped coefficient product (1.7 -> 0.9 us of an Fq12 product's 6.3 us on BN254)
    static MG_DEV F times9(const F &a) {
        u32 t[N + 1];
        u64 c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c += (u64)a.v[i] * 9u;
            t[i] = (u32)c;
            c >>= 32;
        }
        t[N] = (u32)c;
        const u64 top64 = ((u64)t[N] << 32) | t[N - 1];
        const float inv = (1.0f - 1.0f / 1048576.0f) / (float)((u64)K::Fq::P[N - 1] + 1u);
        const u32 q = (u32)((float)top64 * inv);
        F r;
        u64 m = 0;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            m += (u64)q * K::Fq::P[i];
            const u64 d = (u64)t[i] - (u32)m - bw;
            r.v[i] = (u32)d;
            bw = (u32)(d >> 63);
            m >>= 32;
        }
        return F::reduce_once(r, t[N] - (u32)m - bw);
    }
    static MG_DEV F xi_real(const F &a) { // U0 a, xi = U0 + u
        if constexpr (K::U0 == 9) return times9(a);
        else return P::small_mul(a);
    }
    static MG_DEV F2 mul2_xi(const F2 &a) {
        return F2{F::sub(xi_real(a.c0), a.c1), F::add(xi_real(a.c1), a.c0)};
    }
    // A wavefront runs in lockstep and its LDS operations complete in order, so lanes exchange data through LDS with a
    // compiler-level fence only -- no s_barrier. (The Miller kernel runs two wavefronts with different programs in one
    // workgroup; a workgroup barrier here would have to be matched instruction for instruction.)
    static MG_DEV void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    static MG_DEV int slot_of(int m) { return 3 * (m & 1) + (m >> 1); } // exponent of w -> slot inside a register

    // a / 2: the Montgomery representative halves with the value
    static MG_DEV F half(const F &a) {
        u32 t[N];
        const u32 odd = 0u - (a.v[0] & 1u);
        u64 c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c += (u64)a.v[i] + (K::Fq::P[i] & odd);
            t[i] = (u32)c;
            c >>= 32;
        }
        F r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = (t[i] >> 1) | ((i + 1 < N ? t[i + 1] : (u32)c) << 31);
        return r;
    }
    // operand k of a level's product on lane k: every lane runs the same instruction stream (a chain of `if (lane == ..)`
    // branches would execute the products one after another)
    static MG_DEV F2 pick(int k, const F2 &a0, const F2 &a1, const F2 &a2, const F2 &a3, const F2 &a4) {
        return F2::select(k == 0, a0, F2::select(k == 1, a1, F2::select(k == 2, a2, F2::select(k == 3, a3, a4))));
    }
    static MG_DEV F2 half2(const F2 &a) { return F2{half(a.c0), half(a.c1)}; }
    static MG_DEV F2 triple(const F2 &a) { return P::add(P::dbl(a), a); }

    // Twelve lanes, one per Fq component of the result: coefficient m = sum over i of the product in slot 6 i + ((m - i) mod 6),
    // for the columns that were filled (bit j of `cols`). The products that wrap around w^6 = xi were multiplied by xi by the
    // lane that made them, so this is a plain sum of at most six Fq values -- the first version folded on six lanes with the
    // xi multiplication at the end and spent more time here (6.6 us) than in the 36 products (5 us).
    static __device__ __noinline__ void fold(int d, u32 cols) {
        if (lane_id() < 12) {
            const int m = lane_id() >> 1, comp = lane_id() & 1;
            // all six loads first (their LDS latencies overlap), then a three-deep tree of additions; a column that was not
            // filled contributes zero
            F t[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int j = m - i;
                j += j < 0 ? 6 : 0;
                t[i] = F::load(base() + (PROD + 6 * i + j) * W + comp * N);
                if (!((cols >> j) & 1u)) t[i] = F::zero();
            }
            const F acc = F::add(F::add(F::add(t[0], t[1]), F::add(t[2], t[3])), F::add(t[4], t[5]));
            acc.store(base() + (d + slot_of(m)) * W + comp * N);
        }
        sync();
    }
    // d = a * b (registers; d may be a or b)
    static __device__ __noinline__ void mul12(int d, int a, int b) {
        if (lane_id() < 36) {
            const int i = lane_id() / 6, j = lane_id() - 6 * i;
            F2 v = mul2(ld(a + slot_of(i)), ld(b + slot_of(j)));
            if (i + j >= 6) v = mul2_xi(v);
            st(PROD + lane_id(), v);
        }
        sync();
        fold(d, 0x3fu);
    }
    // (sum of c_m w_m) mod p for small per-lane integers c_m = cp_m - cn_m (one of the two zero, either sum below 256), canonical
    // in and out: the two sums as 64-bit columns without carries, t = pos + 256 p - neg in N + 1 words, one quotient estimate from
    // its top 64 bits (never above floor(t / p), at most one below -- as in times9), one conditional subtraction. A chain of
    // modular additions costs ~90 instructions per operand (sum, comparison with p, selection); this is two multiply-adds per
    // operand word and ~150 instructions at the end, and small multiples (3 T - 2 a, xi = 9 + u) come for free.
    template <int M> static MG_DEV F lincomb(const F *w, const u32 *cp, const u32 *cn) {
        u64 pos[N], neg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) pos[i] = 0, neg[i] = 0;
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int i = 0; i < N; ++i) pos[i] += (u64)cp[m] * w[m].v[i], neg[i] += (u64)cn[m] * w[m].v[i];
        u32 t[N + 1], qn[N + 1];
        u64 c = 0, e = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u32 kp = (K::Fq::P[i] << 8) | (i ? K::Fq::P[i - 1] >> 24 : 0u); // word i of 256 p
            c += pos[i] + kp, e += neg[i];
            t[i] = (u32)c, qn[i] = (u32)e;
            c >>= 32, e >>= 32;
        }
        t[N] = (u32)c + (K::Fq::P[N - 1] >> 24), qn[N] = (u32)e;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i <= N; ++i) {
            const u64 d = (u64)t[i] - qn[i] - bw;
            t[i] = (u32)d;
            bw = (u32)(d >> 63);
        }
        const u64 top64 = ((u64)t[N] << 32) | t[N - 1];
        const float inv = (1.0f - 1.0f / 1048576.0f) / (float)((u64)K::Fq::P[N - 1] + 1u);
        const u32 q = (u32)((float)top64 * inv);
        F r;
        u64 mm = 0;
        bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            mm += (u64)q * K::Fq::P[i];
            const u64 d = (u64)t[i] - (u32)mm - bw;
            r.v[i] = (u32)d;
            bw = (u32)(d >> 63);
            mm >>= 32;
        }
        return F::reduce_once(r, t[N] - (u32)mm - bw);
    }
    // d = a^2 (registers; d may be a) with ONE Fq product per lane. With a lone wavefront on its SIMD every instruction costs
    // 5-9 cycles whether or not it depends on the one before (profiles/r02_ubench_pairing.txt), so what counts is the length of
    // the one instruction stream. A square has 21 distinct coefficient products a_i a_j (i <= j): the six squares as
    // P_i = (x + y)(x - y) and Q_i = x y, the fifteen others as Karatsuba's three -- 57 Fq products = 57 lanes, one product
    // deep. Stage B, twelve lanes, one Fq component of the result each: the component is a fixed small-integer combination of
    // at most ten of those products (doubling, Karatsuba's differences and xi = U0 + u where i + j wraps around w^6 all folded
    // into the coefficients, |c| <= 2 U0 + 2), one `lincomb`. Round 3 did this in two stages of modular additions (42 + 12
    // lanes): 5.2 us; now 3.2. Same field elements as mul12(d, a, a), canonical throughout.
    static constexpr u64 SQ_PAIR_I = 0x433222111100000ull, SQ_PAIR_J = 0x554543543254321ull; // (i, j), i < j, a nibble per pair
    static constexpr int SQ_OPS = 10;
    struct SqTab {
        u32 idx[2], cf[3]; // this lane's stage B: ten product slots (6 bits each), ten coefficients (a byte each: magnitude | 0x80 minus)
    };
    static constexpr SqTab sq_entry(int l) {
        const int m = l >> 1, comp = l & 1, k = K::U0;
        u32 slot[SQ_OPS] = {}, cf[SQ_OPS] = {};
        int n = 0, pair = 0;
        constexpr u32 MINUS = 0x80u;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) {
                const int p = pair;
                if (j > i) ++pair;
                if ((i + j) % 6 != m) continue;
                const bool wrap = i + j >= 6;
                if (i == j) { // a_i^2 = P + 2 Q u; times xi: (k P - 2 Q) + (2 k Q + P) u
                    const u32 P = 2 * i, Q = 2 * i + 1;
                    if (!wrap) slot[n] = comp ? Q : P, cf[n] = comp ? 2 : 1, ++n;
                    else if (!comp) slot[n] = P, cf[n] = k, ++n, slot[n] = Q, cf[n] = 2 | MINUS, ++n;
                    else slot[n] = Q, cf[n] = 2 * k, ++n, slot[n] = P, cf[n] = 1, ++n;
                } else { // 2 a_i a_j = 2 (t0 - t1) + 2 (t2 - t0 - t1) u; times xi: ((2k+2) t0 - (2k-2) t1 - 2 t2) + (2k t2 - (2k-2) t0 - (2k+2) t1) u
                    const u32 t0 = 12 + 3 * p, t1 = t0 + 1, t2 = t0 + 2;
                    if (!wrap && !comp) slot[n] = t0, cf[n] = 2, ++n, slot[n] = t1, cf[n] = 2 | MINUS, ++n;
                    else if (!wrap) slot[n] = t2, cf[n] = 2, ++n, slot[n] = t0, cf[n] = 2 | MINUS, ++n, slot[n] = t1, cf[n] = 2 | MINUS, ++n;
                    else if (!comp) {
                        slot[n] = t0, cf[n] = 2 * k + 2, ++n, slot[n] = t2, cf[n] = 2 | MINUS, ++n;
                        if (k > 1) slot[n] = t1, cf[n] = (2 * k - 2) | MINUS, ++n;
                    } else {
                        slot[n] = t2, cf[n] = 2 * k, ++n, slot[n] = t1, cf[n] = (2 * k + 2) | MINUS, ++n;
                        if (k > 1) slot[n] = t0, cf[n] = (2 * k - 2) | MINUS, ++n;
                    }
                }
            }
        SqTab t{{0, 0}, {0, 0, 0}};
        for (int e = 0; e < SQ_OPS; ++e) t.idx[e / 5] |= slot[e] << (6 * (e % 5)), t.cf[e / 4] |= cf[e] << (8 * (e % 4));
        return t;
    }
    static MG_DEV SqTab sq_tab() { // this lane's entry (once per kernel)
        SqTab v{{0, 0}, {0, 0, 0}};
#pragma unroll
        for (int l = 0; l < 12; ++l) {
            const SqTab t = sq_entry(l);
            const bool me = lane_id() == l;
            v.idx[0] = me ? t.idx[0] : v.idx[0], v.idx[1] = me ? t.idx[1] : v.idx[1];
            v.cf[0] = me ? t.cf[0] : v.cf[0], v.cf[1] = me ? t.cf[1] : v.cf[1], v.cf[2] = me ? t.cf[2] : v.cf[2];
        }
        return v;
    }
    static __device__ __noinline__ void sqr12(int d, int a, const SqTab tab) {
        const int l = lane_id();
        u32 *q = base() + PROD * W; // Fq slot t = q + t N (72 of them)
        if (l < 57) {
            const bool sq = l < 12;
            const int p = sq ? 0 : (l - 12) / 3;
            const int k = sq ? (l & 1) : (l - 12) - 3 * p;
            const int i = sq ? (l >> 1) : (int)((SQ_PAIR_I >> (4 * p)) & 15u), j = sq ? i : (int)((SQ_PAIR_J >> (4 * p)) & 15u);
            const F2 ai = ld(a + slot_of(i)), aj = ld(a + slot_of(j));
            const F si = F::add(ai.c0, ai.c1), sj = F::add(aj.c0, aj.c1), di = F::sub(ai.c0, ai.c1);
            // squares: k = 0 (x + y)(x - y), k = 1 x y; pairs: k = 0 x_i x_j, k = 1 y_i y_j, k = 2 (x_i + y_i)(x_j + y_j)
            const F x = sq ? F::select(k == 0, si, ai.c0) : F::select(k == 0, ai.c0, F::select(k == 1, ai.c1, si));
            const F y = sq ? F::select(k == 0, di, ai.c1) : F::select(k == 0, aj.c0, F::select(k == 1, aj.c1, sj));
            F::mul(x, y).store(q + l * N);
        }
        sync();
        if (l < 12) {
            F w[SQ_OPS];
            u32 cp[SQ_OPS], cn[SQ_OPS];
#pragma unroll
            for (int e = 0; e < SQ_OPS; ++e) {
                const u32 cf = (tab.cf[e / 4] >> (8 * (e % 4))) & 255u;
                cp[e] = (cf & 0x80u) ? 0u : cf, cn[e] = (cf & 0x80u) ? (cf & 0x7fu) : 0u;
                w[e] = F::load(q + ((tab.idx[e / 5] >> (6 * (e % 5))) & 63u) * N);
            }
            lincomb<SQ_OPS>(w, cp, cn).store(base() + (d + slot_of(l >> 1)) * W + (l & 1) * N);
        }
        sync();
    }
    struct CycTab {
        u32 idx, clo, chi; // stage B of this lane: six product slots, 5 bits each; seven coefficients (six products and a), a byte each: magnitude | 0x80 minus
    };
    static constexpr CycTab cyc_entry(int l) {
        const int m = l >> 1, c = l & 1;
        u32 w[6] = {0, 0, 0, 0, 0, 0}, k[7] = {0, 0, 0, 0, 0, 0, 0};
        constexpr u32 MINUS = 0x80u, T3 = 3u, XI3 = 3u * (u32)K::U0; // result = 3 T +- 2 a, T = X + U0 Y +- Z
        if (!(m & 1)) { // a_m' = 3 (a_i^2 + xi a_j^2) - 2 a_m, i = m / 2, j = i + 3: P_i + U0 P_j - Q_j | Q_i + U0 Q_j + P_j
            const int i = m >> 1, j = i + 3;
            w[0] = 2 * i + c, k[0] = T3;
            w[2] = 2 * j + c, k[2] = XI3;
            w[4] = 2 * j + (c ^ 1), k[4] = T3 | (c ? 0u : MINUS);
            k[6] = 2u | MINUS;
        } else { // a_3' = 3 (2 a0 a3) + 2 a_3, a_5' = 3 (2 a1 a4) + 2 a_5, a_1' = 3 xi (2 a2 a5) + 2 a_1
            const int base = 12 + 4 * (m == 3 ? 0 : (m == 5 ? 1 : 2));
            const int o = m == 1 ? 2 : 0; // a_1' takes xi t5: U0 times this component of the product, +- the other one
            w[o] = base + 2 * c, k[o] = m == 1 ? XI3 : T3;
            w[o + 1] = base + 2 * c + 1, k[o + 1] = (m == 1 ? XI3 : T3) | (c ? 0u : MINUS); // c0 = U0' - U1', c1 = U2' + U3'
            if (m == 1) {
                w[4] = base + 2 * (c ^ 1), k[4] = T3 | (c ? 0u : MINUS);
                w[5] = base + 2 * (c ^ 1) + 1, k[5] = T3 | MINUS; // c0: -(U2' + U3'); c1: + (U0' - U1')
            }
            k[6] = 2u;
        }
        return CycTab{w[0] | w[1] << 5 | w[2] << 10 | w[3] << 15 | w[4] << 20 | w[5] << 25, k[0] | k[1] << 8 | k[2] << 16 | k[3] << 24,
                      k[4] | k[5] << 8 | k[6] << 16};
    }
    static MG_DEV CycTab cyc_tab() { // this lane's entry (once per kernel)
        CycTab v{0u, 0u, 0u};
#pragma unroll
        for (int l = 0; l < 12; ++l) {
            const CycTab t = cyc_entry(l);
            const bool me = lane_id() == l;
            v.idx = me ? t.idx : v.idx, v.clo = me ? t.clo : v.clo, v.chi = me ? t.chi : v.chi;
        }
        return v;
    }
    static __device__ __noinline__ void cyc_sqr12(int d, int a, const CycTab tab) {
        const int l = lane_id();
        u32 *q = base() + PROD * W;
        if (l < 24) {
            const bool isq = l < 12;
            const int i = isq ? (l >> 1) : ((l - 12) >> 2), kk = isq ? (l & 1) : ((l - 12) & 3);
            const F2 ai = ld(a + slot_of(i)), aj = ld(a + slot_of(isq ? i : i + 3));
            const F s1 = F::select(isq || !(kk & 1), ai.c0, ai.c1), s2 = F::select(isq, F::select(kk != 0, ai.c0, ai.c1), s1);
            const F vs = F::select(isq, ai.c1, F::select(kk == 0 || kk == 3, aj.c0, aj.c1));
            F::mul(F::add(s1, s2), F::select(isq && kk == 0, F::sub(ai.c0, ai.c1), vs)).store(q + l * N);
        }
        sync();
        if (l < 12) {
            F w[7];
            u32 cp[7], cn[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const u32 cf = ((k < 4 ? tab.clo : tab.chi) >> (8 * (k & 3))) & 255u;
                cp[k] = (cf & 0x80u) ? 0u : cf, cn[k] = (cf & 0x80u) ? (cf & 0x7fu) : 0u;
                w[k] = k < 6 ? F::load(q + ((tab.idx >> (5 * k)) & 31u) * N) : F::load(base() + (a + slot_of(l >> 1)) * W + (l & 1) * N);
            }
            lincomb<7>(w, cp, cn).store(base() + (d + slot_of(l >> 1)) * W + (l & 1) * N);
        }
        sync();
    }
    // f *= line(P), arkworks `ell`: D-type twist (BN254) c0 py + (c1 px) w + c2 w^3 (mul_by_034), M-type (BLS12-381)
    // c0 + (c1 px) w^2 + (c2 py) w^3 (mul_by_014). Round 4: ONE Fq PRODUCT PER LANE. The line arrives as a ring entry made by
    // wave 1 (below): the three coefficients L_t already scaled by px / py -- so the eighteen Fq2 products f_i L_t are 54
    // Karatsuba products on 54 lanes (one product deep), and each Fq component of the result is one `lincomb` of the nine
    // products behind it (Karatsuba's differences and xi, where i + LINE_J(t) wraps around w^6, in the coefficients): 1.9 us
    // against the 7.9 of the first version (an Fq2-by-Fq product, an Fq2 product and the xi multiple, all on the lane of one
    // coefficient product, then a fold).
    static constexpr int LINE_J(int t) { return K::TWIST_D ? (t == 2 ? 3 : t) : (t == 0 ? 0 : t + 1); } // L_t sits at w^LINE_J(t)
    static constexpr int RINGW = 3 * W; // ring entry: L_0, L_1, L_2
    struct EllTab {
        u32 idx[2], cf[3]; // this lane's second stage: nine product slots (6 bits each) and coefficients (a byte each: magnitude | 0x80 minus)
    };
    static constexpr EllTab ell_entry(int l) {
        const int m = l >> 1, comp = l & 1, k = K::U0;
        constexpr u32 MINUS = 0x80u;
        EllTab r{{0, 0}, {0, 0, 0}};
        for (int t = 0; t < 3; ++t) {
            int i = m - LINE_J(t);
            const bool wrap = i < 0;
            i += wrap ? 6 : 0;
            const u32 t0 = 3 * (3 * i + t), c0 = !wrap ? (comp ? 1 | MINUS : 1) : (comp ? (k - 1) | MINUS : k + 1),
                      c1 = !wrap ? 1 | MINUS : (comp ? (k + 1) | MINUS : (k - 1) | MINUS), c2 = !wrap ? (comp ? 1 : 0) : (comp ? k : 1 | MINUS);
            const u32 sl[3] = {t0, t0 + 1, t0 + 2}, cf[3] = {c0, c1, c2}; // v = (t0 - t1) + (t2 - t0 - t1) u; xi v = ((k+1) t0 - (k-1) t1 - t2) + (k t2 - (k-1) t0 - (k+1) t1) u
            for (int e = 0; e < 3; ++e) {
                const int o = 3 * t + e;
                r.idx[o / 5] |= sl[e] << (6 * (o % 5)), r.cf[o / 4] |= ((cf[e] & 0x7fu) ? cf[e] : 0u) << (8 * (o % 4));
            }
        }
        return r;
    }
    static MG_DEV EllTab ell_tab() { // this lane's entry (once per kernel)
        EllTab v{{0, 0}, {0, 0, 0}};
#pragma unroll
        for (int l = 0; l < 12; ++l) {
            const EllTab t = ell_entry(l);
            const bool me = lane_id() == l;
            v.idx[0] = me ? t.idx[0] : v.idx[0], v.idx[1] = me ? t.idx[1] : v.idx[1];
            v.cf[0] = me ? t.cf[0] : v.cf[0], v.cf[1] = me ? t.cf[1] : v.cf[1], v.cf[2] = me ? t.cf[2] : v.cf[2];
        }
        return v;
    }
    static __device__ __noinline__ void ell(int f, int o, const EllTab tab) {
        const int l = lane_id();
        const u32 *e = ring_slot(o);
        u32 *q = base() + PROD * W; // Fq slot s = q + s N (72 of them)
        if (l < 54) {
            const int pr = l / 3, k = l - 3 * pr, i = pr / 3, t = pr - 3 * i;
            const F2 fi = ld(f + slot_of(i)), lt = F2::load(e + t * W);
            const F x = F::select(k == 0, fi.c0, F::select(k == 1, fi.c1, F::add(fi.c0, fi.c1)));
            const F y = F::select(k == 0, lt.c0, F::select(k == 1, lt.c1, F::add(lt.c0, lt.c1)));
            F::mul(x, y).store(q + l * N);
        }
        sync();
        if (l < 12) {
            F w[9];
            u32 cp[9], cn[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const u32 cf = (tab.cf[c / 4] >> (8 * (c % 4))) & 255u;
                cp[c] = (cf & 0x80u) ? 0u : cf, cn[c] = (cf & 0x80u) ? (cf & 0x7fu) : 0u;
                w[c] = F::load(q + ((tab.idx[c / 5] >> (6 * (c % 5))) & 63u) * N);
            }
            lincomb<9>(w, cp, cn).store(base() + (f + slot_of(l >> 1)) * W + (l & 1) * N);
        }
        sync();
    }
    static MG_DEV void set_one(int d) {
        if (lane_id() < 6) st(d + lane_id(), lane_id() == 0 ? F2::one() : F2::zero());
        sync();
    }
    static MG_DEV void copy12(int d, int a) {
        if (lane_id() < 6) st(d + lane_id(), ld(a + lane_id()));
        sync();
    }
    static MG_DEV void conj12(int d) { // the p^6-power Frobenius: odd powers of w change sign
        if (lane_id() >= 3 && lane_id() < 6) st(d + lane_id(), P::neg(ld(d + lane_id())));
        sync();
    }
    // x -> x^(q^k), in place
    template <int k> static __device__ __noinline__ void frob12(int d) {
        if (lane_id() < 6) {
            const int i = lane_id() / 3, j = lane_id() - 3 * i;
            F2 v = ld(d + lane_id());
            if (k & 1) v = P::conj(v);
            const auto &ca = k == 1 ? K::FROB6A_1 : (k == 2 ? K::FROB6A_2 : K::FROB6A_3);
            const auto &cb = k == 1 ? K::FROB6B_1 : (k == 2 ? K::FROB6B_2 : K::FROB6B_3);
            const auto &cg = k == 1 ? K::FROB12_1 : (k == 2 ? K::FROB12_2 : K::FROB12_3);
            if (j) v = mul2(v, F2::select(j == 1, P::f2const(ca), P::f2const(cb)));
            if (i) v = mul2(v, P::f2const(cg));
            st(d + lane_id(), v);
        }
        sync();
    }
    // 1/a in Fq on one lane: Kaliski's almost-inverse on the Montgomery representative y = a R (an integer below p) -- u, v shrink
    // by a bit per step while r, s only double and add (plain integers below 2p: no halving modulo p inside the loop, which was
    // two thirds of the binary extended Euclid of round 2: 127 us) -- gives y^-1 2^k, N' <= k <= 2 N' (N' = bits of p); four
    // Montgomery products then turn a^-1 R^-1 2^k into a^-1 R: two by R^2, two by the plain integers 2^(32N - j) that take
    // j <= 32N bits of the 2^k off each. 94 us (the same loop on the scalar unit, its input read with readfirstlane: 114). a != 0 (an Fq12 norm of a non-zero element).
    static __device__ __noinline__ F inv_euclid(const F &a) {
        F u, v = a, r = F::zero(), sv = F::zero();
#pragma unroll
        for (int i = 0; i < N; ++i) u.v[i] = K::Fq::P[i];
        sv.v[0] = 1;
        auto shr1 = [](F &t) {
#pragma unroll
            for (int i = 0; i < N; ++i) t.v[i] = (t.v[i] >> 1) | ((i + 1 < N ? t.v[i + 1] : 0u) << 31);
        };
        auto shl1 = [](F &t) {
#pragma unroll
            for (int i = N - 1; i >= 0; --i) t.v[i] = (t.v[i] << 1) | (i ? t.v[i - 1] >> 31 : 0u);
        };
        auto gt = [](const F &x, const F &y) { // x > y
            bool g = false;
#pragma unroll
            for (int i = 0; i < N; ++i) g = x.v[i] != y.v[i] ? x.v[i] > y.v[i] : g;
            return g;
        };
        auto isub = [](F &x, const F &y) { // plain x -= y (x >= y)
            u32 bw = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const u64 dd = (u64)x.v[i] - y.v[i] - bw;
                x.v[i] = (u32)dd;
                bw = (u32)(dd >> 63);
            }
        };
        auto iadd = [](F &x, const F &y) { // plain x += y (the sum stays below 2p < 2^(32 N))
            u32 c = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const u64 ss = (u64)x.v[i] + y.v[i] + c;
                x.v[i] = (u32)ss;
                c = (u32)(ss >> 32);
            }
        };
        int k = 0;
        while (!v.is_zero()) { // invariants: y r = -u 2^k, y s = v 2^k (mod p); gcd(u, v) = 1
            if (!(u.v[0] & 1u)) shr1(u), shl1(sv);
            else if (!(v.v[0] & 1u)) shr1(v), shl1(r);
            else if (gt(u, v)) isub(u, v), shr1(u), iadd(r, sv), shl1(sv);
            else isub(v, u), shr1(v), iadd(sv, r), shl1(r);
            ++k;
        }
        F pp;
#pragma unroll
        for (int i = 0; i < N; ++i) pp.v[i] = K::Fq::P[i];
        if (!gt(pp, r)) isub(r, pp); // r < 2p
        isub(pp, r);                 // y^-1 2^k = p - r
        F r2;
#pragma unroll
        for (int i = 0; i < N; ++i) r2.v[i] = K::Fq::R2[i];
        F t = F::mul(F::mul(pp, r2), r2); // a^-1 2^k R
        auto pow2 = [](int e) {            // the integer 2^e, e < 32 N
            F z = F::zero();
#pragma unroll
            for (int i = 0; i < N; ++i) z.v[i] = (e >> 5) == i ? 1u << (e & 31) : 0u;
            return z;
        };
        const int j1 = k < 32 * N ? k : 32 * N - 1, j2 = k - j1; // k <= 2 (32 N - 2): both below 32 N, j1 >= 1
        t = F::mul(t, pow2(32 * N - j1));
        if (j2) t = F::mul(t, pow2(32 * N - j2));
        return t;
    }
    // d = 1 / a (registers; t1, t2 scratch; d, t1, t2 distinct from a): 1/a = conj(a) / (a conj(a)), and a conj(a) lies in
    // Fq6 (its odd coefficients cancel exactly), whose inverse is ark-ff's Fp6 formula with the Fq2 products of each level
    // on different lanes; the one Fq inversion at the bottom stays on lane 0. (The first version ran the whole tower
    // inversion on lane 0 through pairing_dev.h's out-of-line helpers: 0.38 ms of a 4.6 ms verification.)
    static __device__ __noinline__ void inv12(int d, int a, int t1, int t2) {
        copy12(t1, a);
        conj12(t1);
        mul12(t2, a, t1); // (n0, n1, n2, 0, 0, 0)
        const F2 n0 = ld(t2), n1 = ld(t2 + 1), n2 = ld(t2 + 2);
        const int l = lane_id();
        if (l < 6) // n0^2, n1 n2, n2^2, n0 n1, n1^2, n0 n2
            st(PROD + l, mul2(l == 5 ? n0 : pick(l, n0, n1, n2, n0, n1), l == 5 ? n2 : pick(l, n0, n2, n2, n1, n1)));
        sync();
        const F2 s0 = P::sub(ld(PROD), mul2_xi(ld(PROD + 1))), s1 = P::sub(mul2_xi(ld(PROD + 2)), ld(PROD + 3)),
                 s2 = P::sub(ld(PROD + 4), ld(PROD + 5));
        sync();
        if (l < 3) st(PROD + l, mul2(pick(l, n0, n2, n1, n1, n1), pick(l, s0, s1, s2, s2, s2)));
        sync();
        const F2 dd = P::add(ld(PROD), mul2_xi(P::add(ld(PROD + 1), ld(PROD + 2)))); // the norm down to Fq2
        sync();
        if (l == 0) {
            const F x[2] = {dd.c0, dd.c1};
            F y[2];
            F::template mul_many<2>(x, x, y);
            st(PROD, F2{inv_euclid(F::add(y[0], y[1])), F::zero()});
        }
        sync();
        const F2 di = mul2_fp(F2{dd.c0, F::neg(dd.c1)}, ld(PROD).c0); // 1 / dd
        sync();
        if (l < 3) st(t2 + l, mul2(pick(l, s0, s1, s2, s2, s2), di)); // t2 = 1 / (a conj(a)); its slots 3..5 are zero already
        sync();
        mul12(d, t1, t2);
    }
    // |x| in width-4 non-adjacent form: odd digits in (-8, 8), at most one in any four positions -- BN254's 63-bit x has 14
    // of them against 28 one bits, and an inverse in the cyclotomic subgroup is a conjugation
    struct WNaf {
        u64 nz = 0, neg = 0, m0 = 0, m1 = 0; // digit i: non-zero, negative, magnitude 1 + 2 (m0 + 2 m1)
        int top = 0;
    };
    static constexpr WNaf wnaf4(u64 x) {
        WNaf r;
        unsigned __int128 v = x;
        for (int i = 0; v; ++i, v >>= 1)
            if (v & 1) {
                int dgt = (int)(v & 15);
                if (dgt >= 8) dgt -= 16;
                v -= dgt; // (v is odd and dgt has its sign: no wrap)
                const int mag = (dgt < 0 ? -dgt : dgt) >> 1;
                r.nz |= 1ull << i, r.neg |= (u64)(dgt < 0) << i, r.m0 |= (u64)(mag & 1) << i, r.m1 |= (u64)(mag >> 1) << i, r.top = i;
            }
        return r;
    }
    static constexpr int popcount64(u64 x) { return x ? 1 + popcount64(x & (x - 1)) : 0; }
    static constexpr bool X_WINDOW = popcount64(K::X) > 12 && K::X < (1ull << 63); // (BLS12-381's x has six one bits: plain binary)
    // d = a^|x| for a in the cyclotomic subgroup (d != a; t .. t + 2: three scratch registers for a^3, a^5, a^7)
    static __device__ __noinline__ void pow_x(int d, int a, int t, const CycTab tab) {
        if constexpr (X_WINDOW) {
            constexpr WNaf w = wnaf4(K::X);
            auto odd = [&](int k) { return k == 0 ? a : t + 6 * (k - 1); }; // a^(2 k + 1)
            cyc_sqr12(d, a, tab);
            mul12(odd(1), a, d), mul12(odd(2), odd(1), d), mul12(odd(3), odd(2), d);
            copy12(d, odd((int)((w.m0 >> w.top) & 1) + 2 * (int)((w.m1 >> w.top) & 1))); // (the leading digit is positive)
#pragma unroll 1
            for (int i = w.top - 1; i >= 0; --i) {
                cyc_sqr12(d, d, tab);
                if ((w.nz >> i) & 1) {
                    const int src = odd((int)((w.m0 >> i) & 1) + 2 * (int)((w.m1 >> i) & 1));
                    const bool inv = (w.neg >> i) & 1;
                    if (inv) conj12(src);
                    mul12(d, d, src);
                    if (inv) conj12(src);
                }
            }
        } else {
            copy12(d, a);
            int top = 63;
            while (!((K::X >> top) & 1)) --top;
#pragma unroll 1
            for (int i = top - 1; i >= 0; --i) {
                cyc_sqr12(d, d, tab);
                if ((K::X >> i) & 1) mul12(d, d, a);
            }
        }
    }
    static MG_DEV void exp_by_neg_x(int d, int a, const CycTab tab) {
        pow_x(d, a, R(FINAL_EXP_REGS - 3), tab);
        if constexpr (!K::X_NEG) conj12(d);
    }
    static MG_DEV void exp_by_x(int d, int a, const CycTab tab) {
        pow_x(d, a, R(FINAL_EXP_REGS - 3), tab);
        if constexpr (K::X_NEG) conj12(d);
    }
    static constexpr int FINAL_EXP_REGS = 22; // 0 .. 18: the sequence below; 19 .. 21: pow_x's odd powers
    // register 0 := final_exponentiation(register 0); registers 1..21 are scratch. The sequence is pairing_dev.h's
    // final_exp (ark-ec 0.3 models/{bn,bls12}/mod.rs); squarings of the hard part in the cyclotomic subgroup (cyc_sqr12).
    static __device__ void final_exp() {
        const int f = R(0), f1 = R(1), f2 = R(2), r = R(3);
        const CycTab ct = cyc_tab();
        copy12(f1, f);
        conj12(f1);
        inv12(f2, f, R(4), R(5));
        mul12(r, f1, f2);
        copy12(f2, r);
        frob12<2>(r);
        mul12(r, r, f2);
        if constexpr (K::BN) {
            const int y0 = R(4), y1 = R(5), y2 = R(6), y3 = R(7), y4 = R(8), y5 = R(9), y6 = R(10), y7 = R(11), y8 = R(12),
                      y9 = R(13), y10 = R(14), y11 = R(15), y12 = R(16), y13 = R(17), y14 = R(18), y15 = R(1);
            exp_by_neg_x(y0, r, ct);
            cyc_sqr12(y1, y0, ct);
            cyc_sqr12(y2, y1, ct);
            mul12(y3, y2, y1);
            exp_by_neg_x(y4, y3, ct);
            cyc_sqr12(y5, y4, ct);
            exp_by_neg_x(y6, y5, ct);
            conj12(y3);
            conj12(y6);
            mul12(y7, y6, y4);
            mul12(y8, y7, y3);
            mul12(y9, y8, y1);
            mul12(y10, y8, y4);
            mul12(y11, y10, r);
            copy12(y12, y9);
            frob12<1>(y12);
            mul12(y13, y12, y11);
            frob12<2>(y8);
            mul12(y14, y8, y13);
            conj12(r);
            mul12(y15, r, y9);
            frob12<3>(y15);
            mul12(f, y15, y14);
        } else {
            const int y0 = R(4), y1 = R(5), y2 = R(6), y3 = R(7), y4 = R(8), y5 = R(9);
            cyc_sqr12(y0, r, ct);
            conj12(y0);
            exp_by_x(y5, r, ct);
            cyc_sqr12(y1, y5, ct);
            mul12(y3, y0, y5);
            exp_by_x(y0, y3, ct);
            exp_by_x(y2, y0, ct);
            exp_by_x(y4, y2, ct);
            mul12(y4, y4, y1);
            exp_by_x(y1, y4, ct);
            conj12(y3);
            mul12(y1, y1, y3);
            mul12(y1, y1, r);
            copy12(y3, r);
            conj12(y3);
            mul12(y0, y0, r);
            frob12<3>(y0);
            mul12(y4, y4, y3);
            frob12<1>(y4);
            mul12(y5, y5, y2);
            frob12<2>(y5);
            mul12(y5, y5, y0);
            mul12(y5, y5, y4);
            mul12(f, y5, y1);
        }
    }
    static MG_DEV signed char loop_digit(int i) {
        signed char dgt = 0;
#pragma unroll
        for (int k = 0; k < K::LOOP_LEN; ++k) dgt = (k == i) ? K::LOOP[k] : dgt;
        return dgt;
    }
    // ---- the ring through which wave 1 of the Miller kernel hands lines to wave 0. An entry is what `ell` multiplies with
    // (RINGW words: the coefficients scaled by px / py); wave 1 makes it either from G2Prepared::from(Q) as that
    // runs (prepare<true>, below) or from a stored coefficient table (scale_stored)
    static constexpr int PREP_WORDS_ = 45 * W; // = PREP_SLOTS * W (the enum is declared further down)
    static constexpr int RING_OFF = MILLER_WORDS + PREP_WORDS_, CNT_OFF = RING_OFF + RING * RINGW;
    static constexpr size_t miller_lds_bytes() { return (size_t)(CNT_OFF + 2) * 4; }
    static MG_DEV volatile u32 *counters() { return (volatile u32 *)(mg_pairing_lds + CNT_OFF); } // [0] produced, [1] consumed
    static MG_DEV u32 *ring_slot(int o) { return mg_pairing_lds + RING_OFF + (o % RING) * RINGW; }
    static MG_DEV const u32 *ring_acquire(int o) { // consumer: entry number o is complete
        while ((int)counters()[0] <= o) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return ring_slot(o);
    }
    static MG_DEV void ring_release(int o) { // consumer: done with entries 0..o
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane_id() == 0) counters()[1] = (u32)(o + 1);
    }
    static MG_DEV u32 *ring_reserve(int o) { // producer: the slot of entry o is free again
        while (o - (int)counters()[1] >= RING) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return ring_slot(o);
    }
    static MG_DEV void ring_publish(int o) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane_id() == 0) counters()[0] = (u32)(o + 1);
    }
    // register f := Miller loop of ONE pair; the NCOEFF lines come through the ring in the order of G2Prepared's table
    static __device__ void miller(int f) {
        const SqTab st = sq_tab();
        const EllTab et = ell_tab();
        set_one(f);
        int o = 0;
        auto line = [&]() {
            ring_acquire(o);
            ell(f, o, et);
            ring_release(o);
            ++o;
        };
#pragma unroll 1
        for (int i = K::LOOP_LEN - 2; i >= 0; --i) {
            if (i != K::LOOP_LEN - 2) sqr12(f, f, st);
            line();
            if (loop_digit(i) != 0) line();
        }
        if constexpr (K::BN) {
            line();
            line();
        }
        if constexpr (K::X_NEG) conj12(f);
    }
    // wave 1, Q prepared in advance (the verifying key's -gamma, -delta; arkworks' table of NCOEFF triples in global memory):
    // scale each triple by P's coordinates and hand it on -- well ahead of wave 0, which so never waits
    // for a global load. (px, py, pz) = (x, y, 1) for an affine P; for P = (X, Y, ZZ, ZZZ) the line is taken times ZZ ZZZ --
    // (X ZZZ, Y ZZ, ZZ ZZZ) -- a factor in Fq, which the final exponentiation sends to 1: no inversion for a point that comes
    // out of a scalar multiplication
    static constexpr int T_PLAIN = K::TWIST_D ? 2 : 0, T_PY = K::TWIST_D ? 0 : 2; // t = 1 is scaled by px on both twists
    static __device__ void scale_stored(const u32 *co, const F &px, const F &py, const F &pz) {
        const int l = lane_id();
#pragma unroll 1
        for (int o = 0; o < P::NCOEFF; ++o) {
            F raw = F::zero();
            const int t = l >> 1, comp = l & 1;
            if (l < 6) raw = F::load(co + (size_t)o * P::COEFFW + t * W + comp * N);
            u32 *e = ring_reserve(o);
            if (l < 6) F::mul(raw, F::select(t == T_PLAIN, pz, F::select(t == T_PY, py, px))).store(e + t * W + comp * N);
            ring_publish(o);
        }
    }
    // ---- G2Prepared::from(Q) on one wavefront, ONE Fq PRODUCT PER LANE (round 4). The first version gave every independent
    // Fq2 product of a doubling / addition step (ark-ec 0.3 models/bn/g2.rs, bls12/g2.rs: doubling_step, addition_step) a lane
    // of its own: three / four Fq2 products deep, 2.95 us each, 18 us per step on average -- the chain of wave 1, not wave 0's
    // Miller loop, bounded a verification's e(A, B) (1.54 against 1.1 ms). Here an Fq2 product is four Fq products on four
    // lanes (a0 b0, a1 b1, a0 b1, a1 b0: 0.94 us) and every linear combination between two product levels -- the
    // recombination c0 = t0 - t1, c1 = t2 + t3 included -- is ONE stage of signed sums of at most four Fq operands, each on
    // the lane of one result component. Both kinds of stage are table driven: a lane's operand slots, signs and destination
    // for every stage are worked out once per kernel (compile-time tables, selected by lane number) and stay in registers, so
    // all lanes run one instruction stream. The formulas are rearranged so that no stage needs more than four operands
    // (same field elements, canonical throughout -- the committed keys stay the known-answer test):
    //   doubling: XY, B = y^2, C = z^2, J = x^2, YZ | E' = -(3 b') C, F = (9 b') C | M = (B - F) / 2, G = (B + F) / 2, H = 2 YZ,
    //             coefficients i = -E' - B, 3 J, -H | x3 = XY M, G^2, E' F = -3 e^2, z3 = B H | y3 = G^2 + E' F
    //   addition: qy z, qx z | theta, lambda | theta^2, lambda^2, theta qx, lambda qy | c, d, j = theta qx - lambda qy, -theta |
    //             e = lambda d, f = z c, g = x d | e, f, g, 2g | h = e + f - 2g, g - h = g + 2g - e - f |
    //             lambda h, theta (g - h), e y, z e | x3, y3, z3        (-Q: the signs of the qy terms flip, no other change)
    enum { SX = 0, SY, SZ, XY, BB, CC, JJ, YZ, NE, FV, MM, GG, HH, CB3, CB9, ZERO, QX, QY, TH, LA, AC, AD, AE, AF, AG, AG2, AH, GH,
           CR0, CR1, CR2, // line coefficients before their scaling by px / py
           PXY,           // (px, py)
           PZ,            // (pz, 0): the factor of the coefficient that px / py do not touch (1 for an affine P)
           QP,            // QP .. QP + 11 hold the (up to twenty-four) Fq products of a level
           PREP_SLOTS = QP + 12 };
    static constexpr size_t prep_lds_bytes() { return (size_t)PREP_SLOTS * W * 4; }
    static constexpr int LANES = 24; // lanes that take part in a stage, at most
    struct Lin {
        u32 ops, dst;
    };
    static constexpr u32 NONE = 0xffffffffu, NEG = 0x80u, OUT = 0x80u, HALF = 0x100u, DEFAULT_DST = 0xffu;
    static constexpr u32 fq(int slot, int comp) { return 2u * (u32)slot + (u32)comp; } // Fq slot: component comp of an Fq2 slot
    static constexpr u32 tq(int p, int k) { return fq(QP, 0) + 4u * (u32)p + (u32)k; } // product k of Fq2 product p
    static constexpr u32 ZQ = fq(ZERO, 0);
    static constexpr u32 ops(u32 a, u32 b = ZQ, u32 c = ZQ, u32 d = ZQ) { return a | b << 8 | c << 16 | d << 24; } // a + b + c + d, NEG: minus
    // component c of Fq2 product p as two signed operands: c0 = t0 - t1, c1 = t2 + t3
    static constexpr u32 re0(int p, int c) { return c ? tq(p, 2) : tq(p, 0); }
    static constexpr u32 re1(int p, int c) { return c ? tq(p, 3) : (tq(p, 1) | NEG); }
    struct ProdTab { // a product level: lane l multiplies Fq slots (e & 255) and (e >> 8 & 255); result to QP's slot l, or where e >> 16 says
        u32 e[LANES] = {};
        int n = 0;
        constexpr void put(u32 a, u32 b, u32 dst = DEFAULT_DST) { e[n++] = a | b << 8 | dst << 16; }
        constexpr void product(int sa, int sb) { // Fq2 slot sa x Fq2 slot sb on four lanes: a0 b0, a1 b1, a0 b1, a1 b0
            for (int k = 0; k < 4; ++k) put(fq(sa, k & 1), fq(sb, (k == 1 || k == 2) ? 1 : 0));
        }
        // ring mode: coefficient t (raw in Fq2 slot raw) times px / py / pz, to the ring entry
        constexpr void scale(int t, int raw) {
            for (int c = 0; c < 2; ++c) put(fq(raw, c), t == T_PLAIN ? fq(PZ, 0) : fq(PXY, t == T_PY ? 1 : 0), OUT | fq(t, c));
        }
    };
    struct LinTab { // a linear level: lane l adds up (at most four) signed Fq slots
        Lin e[LANES] = {};
        u32 flip[LANES] = {}; // sign changes of the operands when -Q is added instead of Q
        int n = 0;
        constexpr void put(u32 o, u32 dst, u32 fl = 0) { e[n] = Lin{o, dst}, flip[n] = fl, ++n; }
        constexpr void rec(int p, int dst_slot) { // Fq2 product p put together
            for (int c = 0; c < 2; ++c) put(ops(re0(p, c), re1(p, c)), fq(dst_slot, c));
        }
        // component c of line coefficient t: arkworks' table (raw mode) / the ring entry or the slot its scaling reads (ring mode)
        constexpr void coeff(bool ring, int t, int c, u32 o, u32 fl = 0) {
            if (!ring) put(o, OUT | fq(t, c), fl);
            else put(o, fq(CR0 + t, c), fl);
        }
    };
    static constexpr int T_DBL_NH = K::TWIST_D ? 0 : 2, T_DBL_I = K::TWIST_D ? 2 : 0; // (-h, 3j, i) or (i, 3j, -h)
    static constexpr int T_ADD_LA = K::TWIST_D ? 0 : 2, T_ADD_J = K::TWIST_D ? 2 : 0; // (lambda, -theta, j) or (j, -theta, lambda)
    // -- doubling
    static constexpr ProdTab d_s1() {
        ProdTab t;
        t.product(SX, SY), t.product(SY, SY), t.product(SZ, SZ), t.product(SX, SX), t.product(SY, SZ);
        return t;
    }
    static constexpr LinTab d_r1() {
        LinTab t;
        t.rec(0, XY), t.rec(1, BB), t.rec(2, CC), t.rec(3, JJ), t.rec(4, YZ);
        return t;
    }
    static constexpr ProdTab d_s2() { // E = 3b' C, F = 9b' C
        ProdTab t;
        t.product(CB3, CC), t.product(CB9, CC);
        return t;
    }
    template <bool RG> static constexpr LinTab d_r2() {
        LinTab t;
        t.put(ops(tq(0, 1), tq(0, 0) | NEG), fq(NE, 0)), t.put(ops(ZQ, tq(0, 2) | NEG, tq(0, 3) | NEG), fq(NE, 1)); // -E
        t.rec(1, FV);
        for (int c = 0; c < 2; ++c) {
            t.put(ops(fq(BB, c), re0(1, c) ^ NEG, re1(1, c) ^ NEG), fq(MM, c) | HALF); // (B - F) / 2
            t.put(ops(fq(BB, c), re0(1, c), re1(1, c)), fq(GG, c) | HALF);             // (B + F) / 2
            t.put(ops(fq(YZ, c), fq(YZ, c)), fq(HH, c));                               // h = (y + z)^2 - b - c = 2 y z
            t.coeff(RG, T_DBL_I, c, ops(re0(0, c), re1(0, c), fq(BB, c) | NEG));       // i = E - B
            t.coeff(RG, T_DBL_NH, c, ops(ZQ, fq(YZ, c) | NEG, fq(YZ, c) | NEG));
            t.coeff(RG, 1, c, ops(fq(JJ, c), fq(JJ, c), fq(JJ, c)));
        }
        return t;
    }
    template <bool RG> static constexpr ProdTab d_s3() {
        ProdTab t;
        t.product(XY, MM), t.product(GG, GG), t.product(NE, FV), t.product(BB, HH);
        if (RG) t.scale(T_PY, CR0 + T_PY), t.scale(1, CR1), t.scale(T_PLAIN, CR0 + T_PLAIN);
        return t;
    }
    template <bool RG> static constexpr LinTab d_r3() {
        LinTab t;
        t.rec(0, SX);
        for (int c = 0; c < 2; ++c) t.put(ops(re0(1, c), re1(1, c), re0(2, c), re1(2, c)), fq(SY, c)); // g^2 - 3 e^2
        t.rec(3, SZ);
        return t;
    }
    // -- addition of (QX, +-QY)
    static constexpr ProdTab a_s1() {
        ProdTab t;
        t.product(QY, SZ), t.product(QX, SZ);
        return t;
    }
    template <bool RG> static constexpr LinTab a_r1() {
        LinTab t;
        for (int c = 0; c < 2; ++c) {
            t.put(ops(fq(SY, c), re0(0, c) ^ NEG, re1(0, c) ^ NEG), fq(TH, c), NEG << 8 | NEG << 16); // -Q: theta = y + qy z
            t.put(ops(fq(SX, c), re0(1, c) ^ NEG, re1(1, c) ^ NEG), fq(LA, c));
            if (!RG) t.put(ops(fq(SX, c), re0(1, c) ^ NEG, re1(1, c) ^ NEG), OUT | fq(T_ADD_LA, c));
        }
        return t;
    }
    static constexpr ProdTab a_s2() {
        ProdTab t;
        t.product(TH, TH), t.product(LA, LA), t.product(TH, QX), t.product(LA, QY);
        return t;
    }
    template <bool RG> static constexpr LinTab a_r2() {
        LinTab t;
        t.rec(0, AC), t.rec(1, AD);
        for (int c = 0; c < 2; ++c) {
            t.coeff(RG, T_ADD_J, c, ops(re0(2, c), re1(2, c), re0(3, c) ^ NEG, re1(3, c) ^ NEG), NEG << 16 | NEG << 24); // -Q: + lambda qy
            t.coeff(RG, 1, c, ops(ZQ, fq(TH, c) | NEG));
        }
        return t;
    }
    template <bool RG> static constexpr ProdTab a_s3() {
        ProdTab t;
        t.product(LA, AD), t.product(SZ, AC), t.product(SX, AD);
        if (RG) t.scale(T_PY, LA), t.scale(1, CR1), t.scale(T_PLAIN, CR0 + T_PLAIN); // (lambda is the coefficient that py scales on either twist)
        return t;
    }
    template <bool RG> static constexpr LinTab a_r3a() {
        LinTab t;
        t.rec(0, AE), t.rec(1, AF), t.rec(2, AG);
        for (int c = 0; c < 2; ++c) t.put(ops(re0(2, c), re1(2, c), re0(2, c), re1(2, c)), fq(AG2, c));
        return t;
    }
    static constexpr LinTab a_r3b() {
        LinTab t;
        for (int c = 0; c < 2; ++c) {
            t.put(ops(fq(AE, c), fq(AF, c), fq(AG2, c) | NEG), fq(AH, c));
            t.put(ops(fq(AG, c), fq(AG2, c), fq(AE, c) | NEG, fq(AF, c) | NEG), fq(GH, c));
        }
        return t;
    }
    static constexpr ProdTab a_s4() {
        ProdTab t;
        t.product(LA, AH), t.product(TH, GH), t.product(AE, SY), t.product(SZ, AE);
        return t;
    }
    static constexpr LinTab a_r4() {
        LinTab t;
        t.rec(0, SX);
        for (int c = 0; c < 2; ++c) t.put(ops(re0(1, c), re1(1, c), re0(2, c) ^ NEG, re1(2, c) ^ NEG), fq(SY, c));
        t.rec(3, SZ);
        return t;
    }
    static_assert(T_ADD_LA == T_PY, "lambda is the coefficient scaled by py");
    // this lane's entry of a table (worked out once per kernel: a chain of selects over compile-time values)
    template <class Tab> static MG_DEV u32 lane_prod(const Tab t) {
        u32 v = NONE;
#pragma unroll
        for (int l = 0; l < LANES; ++l) v = (l < t.n && lane_id() == l) ? t.e[l] : v;
        return v;
    }
    static MG_DEV Lin lane_lin(const LinTab t) {
        Lin v{0u, NONE};
#pragma unroll
        for (int l = 0; l < LANES; ++l) {
            const bool me = l < t.n && lane_id() == l;
            v.ops = me ? t.e[l].ops : v.ops, v.dst = me ? t.e[l].dst : v.dst;
        }
        return v;
    }
    static MG_DEV u32 lane_flip(const LinTab t) {
        u32 v = 0;
#pragma unroll
        for (int l = 0; l < LANES; ++l) v = (l < t.n && lane_id() == l) ? t.flip[l] : v;
        return v;
    }
    static MG_DEV F ldq(u32 q) { return F::load(base() + q * N); }
    // u + v or u - v (canonical in and out): p - v stands in for -v (p itself when v = 0; u + p comes back to u)
    static MG_DEV F addsub(const F &u, const F &v, bool neg) {
        F w;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u64 d = (u64)K::Fq::P[i] - v.v[i] - bw;
            w.v[i] = neg ? (u32)d : v.v[i];
            bw = (u32)(d >> 63);
        }
        return F::add(u, w);
    }
    static MG_DEV u32 *dst_of(u32 d, u32 *out) { return (d & OUT) ? out + (d & 127u) * N : base() + (d & 127u) * N; }
    static MG_DEV void prod_stage(u32 e, u32 *out) {
        if (e != NONE) {
            const u32 d = (e >> 16) & 255u;
            F::mul(ldq(e & 255u), ldq((e >> 8) & 255u)).store(d == DEFAULT_DST ? base() + (fq(QP, 0) + (u32)lane_id()) * N : dst_of(d, out));
        }
        sync();
    }
    template <int NOPS, bool MAY_HALVE = false> static MG_DEV void lin_stage(const Lin e, u32 *out) {
        if (e.dst != NONE) {
            F v[NOPS];
#pragma unroll
            for (int k = 0; k < NOPS; ++k) v[k] = ldq((e.ops >> (8 * k)) & 127u);
            F r = v[0];
#pragma unroll
            for (int k = 1; k < NOPS; ++k) r = addsub(r, v[k], ((e.ops >> (8 * k + 7)) & 1u) != 0);
            if constexpr (MAY_HALVE) r = F::select((e.dst & HALF) != 0, half(r), r);
            r.store(dst_of(e.dst, out));
        }
        sync();
    }
    // The NCOEFF line-coefficient triples of Q (affine, not infinity) in the order the Miller loop consumes them: as arkworks'
    // table in global memory (RG = false: g2_prepare_kernel), or as ring entries for wave 0's Miller loop with P = (px, py)
    template <bool RG> static __device__ void prepare(const F2 &qx, const F2 &qy, u32 *out, const F &px, const F &py, const F &pz) {
        static_assert(PREP_SLOTS * W == PREP_WORDS_, "layout");
        static_assert(fq(PREP_SLOTS, 0) <= 128, "seven bits per operand");
        const u32 ds1 = lane_prod(d_s1()), ds2 = lane_prod(d_s2()), ds3 = lane_prod(d_s3<RG>());
        const Lin dr1 = lane_lin(d_r1()), dr2 = lane_lin(d_r2<RG>()), dr3 = lane_lin(d_r3<RG>());
        const u32 as1 = lane_prod(a_s1()), as2 = lane_prod(a_s2()), as3 = lane_prod(a_s3<RG>()), as4 = lane_prod(a_s4());
        const Lin ar1 = lane_lin(a_r1<RG>()), ar2 = lane_lin(a_r2<RG>()), ar3a = lane_lin(a_r3a<RG>()), ar3b = lane_lin(a_r3b()),
                  ar4 = lane_lin(a_r4());
        const u32 f1 = lane_flip(a_r1<RG>()), f2 = lane_flip(a_r2<RG>());
        if (lane_id() == 0) {
            const F2 b3 = triple(P::f2const(K::B2));
            st(SX, qx), st(SY, qy), st(SZ, F2::one()), st(QX, qx), st(QY, qy), st(ZERO, F2::zero()), st(CB3, b3), st(CB9, triple(b3));
            st(PXY, F2{px, py}), st(PZ, F2{pz, F::zero()});
        }
        sync();
        int o = 0;
        auto dst = [&]() { return RG ? ring_reserve(o) : out + (size_t)o * P::COEFFW; };
        auto done = [&]() {
            if constexpr (RG) ring_publish(o);
            ++o;
        };
        auto doubling = [&]() {
            u32 *c = dst();
            prod_stage(ds1, c);
            lin_stage<2>(dr1, c);
            prod_stage(ds2, c);
            lin_stage<3, true>(dr2, c);
            prod_stage(ds3, c);
            lin_stage<4>(dr3, c);
            done();
        };
        auto addition = [&](bool minus) {
            u32 *c = dst();
            prod_stage(as1, c);
            lin_stage<3>(Lin{ar1.ops ^ (minus ? f1 : 0u), ar1.dst}, c);
            prod_stage(as2, c);
            lin_stage<4>(Lin{ar2.ops ^ (minus ? f2 : 0u), ar2.dst}, c);
            prod_stage(as3, c);
            lin_stage<4>(ar3a, c);
            lin_stage<4>(ar3b, c);
            prod_stage(as4, c);
            lin_stage<4>(ar4, c);
            done();
        };
#pragma unroll 1
        for (int i = K::LOOP_LEN - 2; i >= 0; --i) {
            doubling();
            const signed char dgt = loop_digit(i);
            if (dgt != 0) addition(dgt < 0);
        }
        if constexpr (K::BN) { // + pi(Q) - pi^2(Q)
            const F2 tx = P::f2const(K::TWQ_X), ty = P::f2const(K::TWQ_Y);
            const F2 q1x = mul2(P::conj(qx), tx), q1y = mul2(P::conj(qy), ty);
            const F2 q2x = mul2(P::conj(q1x), tx), q2y = P::neg(mul2(P::conj(q1y), ty));
            if (lane_id() == 0) st(QX, q1x), st(QY, q1y);
            sync();
            addition(false);
            if (lane_id() == 0) st(QX, q2x), st(QY, q2y);
            sync();
            addition(false);
        }
    }
};

} // namespace mg
