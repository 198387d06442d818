// hp_wire.cpp -- the wire / on-disk format of ring elements.
//
// hehub has no serialisation at all; a pipeline that keeps ciphertexts on the device still has to get keys and
// ciphertexts into a process once (per node, per job).  The format is deliberately the device layout with a
// header in front, so loading is one validation pass and one host-to-device copy:
//
//   offset  size      field
//   0       8         magic "HEHUBAMD"
//   8       4         version (1)
//   12      4         kind: 1 polynomial, 2 ciphertext, 3 quadratic ciphertext, 4 key-switching key
//   16      4         log2 of the ring degree N
//   20      4         limbs per polynomial
//   24      4         polynomials (1, 2, 3; for a key 2*digits: rgsw[j][half], digit-major)
//   28      4         representation: 0 coefficient, 1 NTT value
//   32      8         scheme scalar: IEEE-754 bits of the CKKS scaling factor, or the BGV plain modulus, or 0
//   40      8*limbs   the moduli
//   ...     8*polys*limbs*N   words, little-endian u64, polynomial-major then limb-major (= include/hehub_amd.h)
//   end-8   8         FNV-1a-64 of every preceding byte
#include "../../include/hehub_amd.h"

#include <cstring>

namespace {
const char MAGIC[8] = {'H', 'E', 'H', 'U', 'B', 'A', 'M', 'D'};
const size_t FIXED = 40;

uint64_t fnv1a64(const unsigned char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
void put32(unsigned char *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (unsigned char)(v >> (8 * i)); }
void put64(unsigned char *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (unsigned char)(v >> (8 * i)); }
uint32_t get32(const unsigned char *p) { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[i] << (8 * i); return v; }
uint64_t get64(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

bool desc_ok(const hp_wire_desc *d) {
    if (!d || d->kind < HP_WIRE_POLY || d->kind > HP_WIRE_KSK) return false;
    if (d->log_dimension < 1 || d->log_dimension > 16 || d->limbs < 1 || d->limbs > 32 || d->polys < 1) return false;
    if (d->kind == HP_WIRE_POLY && d->polys != 1) return false;
    if (d->kind == HP_WIRE_CT && d->polys != 2) return false;
    if (d->kind == HP_WIRE_QUAD_CT && d->polys != 3) return false;
    if (d->kind == HP_WIRE_KSK && (d->polys % 2 != 0 || d->polys > 64)) return false;
    return d->rep_form <= 1;
}
} // namespace

extern "C" {

uint64_t hp_wire_fnv1a64(const void *data, size_t bytes) { return data ? fnv1a64((const unsigned char *)data, bytes) : 0; }

size_t hp_wire_payload_words(const hp_wire_desc *d) {
    return desc_ok(d) ? (size_t)d->polys * d->limbs * ((size_t)1 << d->log_dimension) : 0;
}

size_t hp_wire_bytes(const hp_wire_desc *d) {
    return desc_ok(d) ? FIXED + 8 * (size_t)d->limbs + 8 * hp_wire_payload_words(d) + 8 : 0;
}

int hp_wire_pack(const hp_wire_desc *d, const uint64_t *moduli, const uint64_t *words, void *buf, size_t cap) {
    const size_t total = hp_wire_bytes(d);
    if (total == 0 || !moduli || !words || !buf || cap < total) return HP_EINVAL;
    unsigned char *b = (unsigned char *)buf;
    memcpy(b, MAGIC, 8);
    put32(b + 8, 1);
    put32(b + 12, d->kind);
    put32(b + 16, d->log_dimension);
    put32(b + 20, d->limbs);
    put32(b + 24, d->polys);
    put32(b + 28, d->rep_form);
    put64(b + 32, d->scheme_scalar);
    for (uint32_t k = 0; k < d->limbs; k++) put64(b + FIXED + 8 * k, moduli[k]);
    unsigned char *pay = b + FIXED + 8 * (size_t)d->limbs;
    const size_t nw = hp_wire_payload_words(d);
    for (size_t i = 0; i < nw; i++) put64(pay + 8 * i, words[i]);
    put64(b + total - 8, fnv1a64(b, total - 8));
    return HP_OK;
}

int hp_wire_unpack(const void *buf, size_t len, hp_wire_desc *d, uint64_t *moduli, size_t moduli_cap,
                   size_t *payload_offset) {
    if (!buf || !d || len < FIXED + 16) return HP_EINVAL;
    const unsigned char *b = (const unsigned char *)buf;
    if (memcmp(b, MAGIC, 8) != 0 || get32(b + 8) != 1) return HP_EINVAL;
    d->kind = get32(b + 12);
    d->log_dimension = get32(b + 16);
    d->limbs = get32(b + 20);
    d->polys = get32(b + 24);
    d->rep_form = get32(b + 28);
    d->scheme_scalar = get64(b + 32);
    const size_t total = hp_wire_bytes(d);
    if (total == 0 || total != len) return HP_EINVAL;
    if (get64(b + total - 8) != fnv1a64(b, total - 8)) return HP_EINVAL;
    if (moduli) {
        if (moduli_cap < d->limbs) return HP_EINVAL;
        for (uint32_t k = 0; k < d->limbs; k++) moduli[k] = get64(b + FIXED + 8 * k);
    }
    if (payload_offset) *payload_offset = FIXED + 8 * (size_t)d->limbs;
    return HP_OK;
}

int hp_dev_wire_load(hp_ctx *ctx, const void *buf, size_t len, uint64_t *d_words) {
    if (!ctx || !buf || !d_words) return HP_EINVAL;
    hp_wire_desc d;
    size_t off = 0;
    int rc = hp_wire_unpack(buf, len, &d, nullptr, 0, &off);
    if (rc) return rc;
    // the payload is little-endian u64 in device order: on a little-endian host it is copied as it is
    return hp_memcpy_h2d(ctx, d_words, (const unsigned char *)buf + off, 8 * hp_wire_payload_words(&d));
}

int hp_dev_wire_store(hp_ctx *ctx, const hp_wire_desc *d, const uint64_t *moduli, const uint64_t *d_words, void *buf,
                      size_t cap) {
    if (!ctx || !d || !d_words) return HP_EINVAL;
    const size_t total = hp_wire_bytes(d);
    if (total == 0 || !moduli || !buf || cap < total) return HP_EINVAL;
    unsigned char *b = (unsigned char *)buf;
    const size_t off = FIXED + 8 * (size_t)d->limbs;
    int rc = hp_memcpy_d2h(ctx, b + off, d_words, 8 * hp_wire_payload_words(d));
    if (rc) return rc;
    memcpy(b, MAGIC, 8);
    put32(b + 8, 1);
    put32(b + 12, d->kind);
    put32(b + 16, d->log_dimension);
    put32(b + 20, d->limbs);
    put32(b + 24, d->polys);
    put32(b + 28, d->rep_form);
    put64(b + 32, d->scheme_scalar);
    for (uint32_t k = 0; k < d->limbs; k++) put64(b + FIXED + 8 * k, moduli[k]);
    put64(b + total - 8, fnv1a64(b, total - 8));
    return HP_OK;
}

} // extern "C"
