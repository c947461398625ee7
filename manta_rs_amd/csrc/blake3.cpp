// BLAKE3 (hash mode, 32-byte output), host code: the checksum manta-parameters puts on every key file
// (`manta_parameters::verify` = `blake3::hash(data) == checksum`, manta-parameters/src/lib.rs:173-177, digests in
// manta-parameters/data.checkfile). Written from the BLAKE3 specification: 1 KiB chunks of 64-byte blocks through a 7-round
// compression function (the ChaCha-style quarter round on a 4 x 4 state of 32-bit words, message words permuted between
// rounds), chunk chaining values merged pairwise into a binary tree whose left subtrees are complete, ROOT flag on the last
// compression. Pinned by the digests of the six verifying-key files the reference ships (tests/test_pin.py).
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace mg {
namespace {

const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline void g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx;
    s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my;
    s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 7);
}
// out = the first 8 words of the compression output (the chaining value; with ROOT: the hash)
void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[8]) {
    uint32_t s[16], m[16], t[16];
    for (int i = 0; i < 8; ++i) s[i] = cv[i];
    for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
    s[12] = (uint32_t)counter, s[13] = (uint32_t)(counter >> 32), s[14] = block_len, s[15] = flags;
    std::memcpy(m, block, 64);
    for (int r = 0; r < 7; ++r) {
        g(s, 0, 4, 8, 12, m[0], m[1]);
        g(s, 1, 5, 9, 13, m[2], m[3]);
        g(s, 2, 6, 10, 14, m[4], m[5]);
        g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]);
        g(s, 1, 6, 11, 12, m[10], m[11]);
        g(s, 2, 7, 8, 13, m[12], m[13]);
        g(s, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
        std::memcpy(m, t, 64);
    }
    for (int i = 0; i < 8; ++i) out[i] = s[i] ^ s[i + 8];
}
void load_block(const uint8_t *p, size_t len, uint32_t w[16]) { // little-endian words, zero padded
    uint8_t b[64] = {0};
    std::memcpy(b, p, len);
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
}
// one chunk (<= 1024 bytes): every block but the last is compressed here; the last block is handed back so that the caller
// can add ROOT when this chunk is the whole input
struct Pending {
    uint32_t cv[8], block[16];
    uint64_t counter;
    uint32_t block_len, flags;
};
Pending chunk_state(const uint8_t *p, size_t len, uint64_t index) {
    Pending o;
    std::memcpy(o.cv, IV, 32);
    o.counter = index;
    size_t off = 0;
    uint32_t start = CHUNK_START;
    while (len - off > 64) {
        uint32_t w[16], next[8];
        load_block(p + off, 64, w);
        compress(o.cv, w, index, 64, start, next);
        std::memcpy(o.cv, next, 32);
        start = 0;
        off += 64;
    }
    load_block(p + off, len - off, o.block);
    o.block_len = (uint32_t)(len - off);
    o.flags = start | CHUNK_END;
    return o;
}
Pending parent_state(const uint32_t left[8], const uint32_t right[8]) {
    Pending o;
    std::memcpy(o.cv, IV, 32);
    std::memcpy(o.block, left, 32);
    std::memcpy(o.block + 8, right, 32);
    o.counter = 0, o.block_len = 64, o.flags = PARENT;
    return o;
}
void finish(const Pending &o, uint32_t extra, uint32_t out[8]) { compress(o.cv, o.block, o.counter, o.block_len, o.flags | extra, out); }

} // namespace

void blake3_hash(const uint8_t *data, size_t len, uint8_t out[32]) {
    uint32_t stack[64][8]; // chaining values of the complete left subtrees, smallest on top
    int depth = 0;
    uint64_t chunks = 0; // chunks already pushed
    size_t off = 0;
    while (len - off > 1024) { // a full chunk that is not the last one
        uint32_t cv[8];
        finish(chunk_state(data + off, 1024, chunks), 0, cv);
        ++chunks;
        for (uint64_t t = chunks; (t & 1) == 0; t >>= 1) { // merge the subtrees this chunk completes
            uint32_t merged[8];
            finish(parent_state(stack[--depth], cv), 0, merged);
            std::memcpy(cv, merged, 32);
        }
        std::memcpy(stack[depth++], cv, 32);
        off += 1024;
    }
    Pending last = chunk_state(data + off, len - off, chunks);
    while (depth > 0) {
        uint32_t cv[8];
        finish(last, 0, cv);
        last = parent_state(stack[--depth], cv);
    }
    uint32_t h[8];
    finish(last, ROOT, h);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)h[i], out[4 * i + 1] = (uint8_t)(h[i] >> 8), out[4 * i + 2] = (uint8_t)(h[i] >> 16), out[4 * i + 3] = (uint8_t)(h[i] >> 24);
    }
}

} // namespace mg
