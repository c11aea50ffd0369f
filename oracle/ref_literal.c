/*
 * oracle/ref_literal.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A literal CPU restatement (plain C) of the fqtk `demux` sample-barcode matcher, written from the
 * behaviour of the reference's three library files.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load this; the product path (fqtk_amd/, libfqtk_match.so) never
 * does and has no CPU fallback.
 *
 * Reference followed (paths relative to /root/reference):
 *   src/lib/mod.rs:26-46     IUPAC_MASKS table
 *   src/lib/mod.rs:49-61     encode()
 *   src/lib/mod.rs:85-92     byte_is_nocall(), is_valid_iupac()
 *   src/lib/bitenc.rs:37-43  BitEnc {storage, width, mask, len, usable_bits_per_block}
 *   src/lib/bitenc.rs:114-121,306-322  push(), set_by_addr(), addr()
 *   src/lib/bitenc.rs:432-459  hamming(self, other, max_mismatches)
 *   src/lib/barcode_matching.rs:55-86    BarcodeMatcher::new
 *   src/lib/barcode_matching.rs:89-110   count_mismatches
 *   src/lib/barcode_matching.rs:119-160  assign_internal
 *   src/lib/barcode_matching.rs:165-186  assign (length reject, no-call prefilter, memo cache)
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against every known-answer vector the
 * reference's own tests hold for this path (tests/golden/reference_kat.json; sources listed there).
 * The Rust reference cannot be compiled in this image (no cargo/rustc), so there is no oracle/_ref.
 *
 * Deliberately literal: u32 block layout, per-block early exit with the adaptive cap, Some-only memo
 * cache.  Do not "optimise" this file; speed lives in the HIP path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ---------------------------------------------------------------- encoding (mod.rs) ---------- */

static uint8_t IUPAC_MASKS[256];
static int g_masks_ready = 0;

/* mod.rs:26-46 */
static void init_masks(void) {
    if (g_masks_ready) return;
    memset(IUPAC_MASKS, 0, sizeof IUPAC_MASKS);
    const uint8_t a = 1, c = 2, g = 4, t = 8;
    IUPAC_MASKS['A'] = a;
    IUPAC_MASKS['C'] = c;
    IUPAC_MASKS['G'] = g;
    IUPAC_MASKS['T'] = t;
    IUPAC_MASKS['U'] = t;
    IUPAC_MASKS['M'] = a | c;
    IUPAC_MASKS['R'] = a | g;
    IUPAC_MASKS['W'] = a | t;
    IUPAC_MASKS['S'] = c | g;
    IUPAC_MASKS['Y'] = c | t;
    IUPAC_MASKS['K'] = g | t;
    IUPAC_MASKS['V'] = a | c | g;
    IUPAC_MASKS['H'] = a | c | t;
    IUPAC_MASKS['D'] = a | g | t;
    IUPAC_MASKS['B'] = c | g | t;
    IUPAC_MASKS['N'] = a | c | g | t;
    g_masks_ready = 1;
}

/* mod.rs:85-87 */
static int byte_is_nocall(uint8_t b) { return b == 'N' || b == 'n' || b == '.'; }

/* u8::to_ascii_uppercase */
static uint8_t to_ascii_uppercase(uint8_t b) { return (b >= 'a' && b <= 'z') ? (uint8_t)(b - 32) : b; }

/* mod.rs:90-92 */
int oracle_is_valid_iupac(uint8_t b) {
    init_masks();
    return IUPAC_MASKS[b] != 0 || byte_is_nocall(b);
}

int oracle_byte_is_nocall(uint8_t b) { return byte_is_nocall(b); }

/* ---------------------------------------------------------------- BitEnc (bitenc.rs) --------- */

typedef struct {
    uint32_t *storage;
    size_t nblocks;
    size_t cap;
    size_t width;
    uint32_t mask;
    size_t len;
    size_t usable_bits_per_block;
} BitEnc;

/* bitenc.rs:53-63 (BitEnc::new) */
static void bitenc_init(BitEnc *b, size_t width) {
    b->storage = NULL;
    b->nblocks = 0;
    b->cap = 0;
    b->width = width;
    b->mask = (uint32_t)((1u << width) - 1u);
    b->len = 0;
    b->usable_bits_per_block = 32 - 32 % width;
}

static void bitenc_free(BitEnc *b) {
    free(b->storage);
    b->storage = NULL;
    b->nblocks = b->cap = b->len = 0;
}

/* bitenc.rs:319-322 */
static void bitenc_addr(const BitEnc *b, size_t i, size_t *block, size_t *bit) {
    size_t k = i * b->width;
    *block = k / b->usable_bits_per_block;
    *bit = k % b->usable_bits_per_block;
}

/* bitenc.rs:114-121 + 311-316 */
static void bitenc_push(BitEnc *b, uint8_t value) {
    size_t block, bit;
    bitenc_addr(b, b->len, &block, &bit);
    if (bit == 0) {
        if (b->nblocks == b->cap) {
            b->cap = b->cap ? b->cap * 2 : 4;
            b->storage = (uint32_t *)realloc(b->storage, b->cap * sizeof(uint32_t));
        }
        b->storage[b->nblocks++] = 0;
    }
    uint32_t m = b->mask << bit;
    b->storage[block] |= m;
    b->storage[block] ^= m;
    b->storage[block] |= ((uint32_t)value & b->mask) << bit;
    b->len += 1;
}

static uint8_t bitenc_get(const BitEnc *b, size_t i) {
    size_t block, bit;
    bitenc_addr(b, i, &block, &bit);
    return (uint8_t)((b->storage[block] >> bit) & b->mask);
}

/* bitenc.rs:432-459.  self = observed, other = expected.  Caller has checked len/width equality. */
static uint32_t bitenc_hamming(const BitEnc *self, const BitEnc *other, uint32_t max_mismatches) {
    uint32_t count = 0;
    size_t values_per_block = self->usable_bits_per_block / self->width;
    for (size_t block_index = 0; block_index < self->nblocks; ++block_index) {
        uint32_t block_diff = self->storage[block_index] & ~other->storage[block_index];
        if (block_diff != 0) {
            size_t shift_i = 0;
            for (size_t v = 0; v < values_per_block; ++v) {
                uint32_t block_diff_sub = (block_diff >> shift_i) & self->mask;
                if (block_diff_sub != 0) count += 1;
                shift_i += self->width;
            }
            if (count >= max_mismatches) return max_mismatches;
        }
    }
    return count;
}

/* mod.rs:49-61 */
static void encode(const uint8_t *bases, size_t n, BitEnc *out) {
    init_masks();
    bitenc_init(out, 4);
    for (size_t i = 0; i < n; ++i) {
        uint8_t base = bases[i];
        uint8_t bit;
        if (byte_is_nocall(base)) {
            bit = IUPAC_MASKS['N'];
        } else {
            bit = IUPAC_MASKS[to_ascii_uppercase(base)];
        }
        bitenc_push(out, bit);
    }
}

/* ---- small exported probes so the golden tests can hit encode()/hamming() directly ---------- */

/* Encodes `n` bases; writes one 4-bit value per base into out_vals (unpacked) and the packed u32
 * blocks into out_blocks (caller provides (n+7)/8 words).  Returns number of blocks. */
size_t oracle_encode(const uint8_t *bases, size_t n, uint8_t *out_vals, uint32_t *out_blocks) {
    BitEnc e;
    encode(bases, n, &e);
    for (size_t i = 0; i < n; ++i) out_vals[i] = bitenc_get(&e, i);
    size_t nb = e.nblocks;
    if (out_blocks) memcpy(out_blocks, e.storage, nb * sizeof(uint32_t));
    bitenc_free(&e);
    return nb;
}

/* hamming() over explicit width-4 value vectors (bitenc.rs test_hamming builds these with
 * push_values, not from ASCII).  Returns -1 when lengths differ (the reference asserts). */
int64_t oracle_hamming_vals(const uint8_t *self_vals, size_t n_self, const uint8_t *other_vals,
                            size_t n_other, uint32_t max_mismatches) {
    if (n_self != n_other) return -1;
    BitEnc a, b;
    bitenc_init(&a, 4);
    bitenc_init(&b, 4);
    for (size_t i = 0; i < n_self; ++i) bitenc_push(&a, self_vals[i]);
    for (size_t i = 0; i < n_other; ++i) bitenc_push(&b, other_vals[i]);
    uint32_t r = bitenc_hamming(&a, &b, max_mismatches);
    bitenc_free(&a);
    bitenc_free(&b);
    return (int64_t)r;
}

/* barcode_matching.rs:89-110 with the test helper's cap of 255 (:212-220).
 * Returns the mismatch count, or -1 on length mismatch (the reference panics with the message
 * "Read barcode (..) length (n) differs from expected barcode (..) length (m) for sample .."). */
int oracle_count_mismatches(const uint8_t *observed, size_t n_obs, const uint8_t *expected,
                            size_t n_exp, uint8_t max_mismatches) {
    BitEnc o, e;
    encode(observed, n_obs, &o);
    encode(expected, n_exp, &e);
    int r;
    if (o.len != e.len) {
        r = -1;
    } else {
        r = (int)bitenc_hamming(&o, &e, (uint32_t)max_mismatches);
    }
    bitenc_free(&o);
    bitenc_free(&e);
    return r;
}

/* ---------------------------------------------------------------- memo cache ------------------ */
/* Stand-in for AHashMap<Vec<u8>, BarcodeMatch> (barcode_matching.rs:9-12,44,84).  Pure memoisation:
 * only Some results are inserted (:177-179), keyed on the raw read bytes (case-sensitive). */

typedef struct {
    uint64_t best_match;
    uint8_t best_mismatches;
    uint8_t next_best_mismatches;
} BarcodeMatch;

typedef struct {
    uint8_t *key; /* NULL = empty slot */
    uint32_t klen;
    uint64_t hash;
    BarcodeMatch val;
} CacheSlot;

typedef struct {
    CacheSlot *slots;
    size_t nslots; /* power of two */
    size_t used;
} Cache;

#define STARTING_CACHE_SIZE 1000000 /* barcode_matching.rs:12 */

static uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    return x;
}

static uint64_t hash_bytes(const uint8_t *p, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ULL ^ (uint64_t)n;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = mix64(h ^ w);
        p += 8;
        n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = mix64(h ^ w ^ ((uint64_t)n << 56));
    }
    return mix64(h);
}

static void cache_init(Cache *c, size_t want) {
    size_t n = 16;
    while (n < want * 2) n <<= 1;
    c->slots = (CacheSlot *)calloc(n, sizeof(CacheSlot));
    c->nslots = n;
    c->used = 0;
}

static void cache_free(Cache *c) {
    if (!c->slots) return;
    for (size_t i = 0; i < c->nslots; ++i) free(c->slots[i].key);
    free(c->slots);
    c->slots = NULL;
}

static CacheSlot *cache_find(Cache *c, const uint8_t *k, uint32_t klen, uint64_t h) {
    size_t i = (size_t)h & (c->nslots - 1);
    for (;;) {
        CacheSlot *s = &c->slots[i];
        if (!s->key) return s;
        if (s->hash == h && s->klen == klen && memcmp(s->key, k, klen) == 0) return s;
        i = (i + 1) & (c->nslots - 1);
    }
}

static void cache_grow(Cache *c) {
    Cache n;
    n.nslots = c->nslots * 2;
    n.slots = (CacheSlot *)calloc(n.nslots, sizeof(CacheSlot));
    n.used = c->used;
    for (size_t i = 0; i < c->nslots; ++i) {
        CacheSlot *s = &c->slots[i];
        if (!s->key) continue;
        size_t j = (size_t)s->hash & (n.nslots - 1);
        while (n.slots[j].key) j = (j + 1) & (n.nslots - 1);
        n.slots[j] = *s;
    }
    free(c->slots);
    *c = n;
}

/* ---------------------------------------------------------------- BarcodeMatcher -------------- */

typedef struct {
    size_t n_samples;
    uint8_t **barcodes; /* upper-cased copies (barcode_matching.rs:71) */
    size_t *barcode_len;
    BitEnc *sample_barcodes;
    size_t max_ns_in_barcodes;
    uint8_t max_mismatches;
    uint8_t min_mismatch_delta;
    int use_cache;
    Cache cache;
    uint64_t cache_hits, cache_misses;
} OracleMatcher;

/* barcode_matching.rs:55-86.  Returns NULL (and writes a message) where the reference panics. */
OracleMatcher *oracle_matcher_new(const char *const *barcodes, size_t n_samples, uint8_t max_mismatches,
                                  uint8_t min_mismatch_delta, int use_cache, char *err, size_t errlen) {
    init_masks();
    if (n_samples == 0) {
        if (err) snprintf(err, errlen, "Must provide at least one sample");
        return NULL;
    }
    for (size_t i = 0; i < n_samples; ++i) {
        if (barcodes[i] == NULL || barcodes[i][0] == '\0') {
            if (err) snprintf(err, errlen, "Sample barcode cannot be empty string");
            return NULL;
        }
    }
    OracleMatcher *m = (OracleMatcher *)calloc(1, sizeof *m);
    m->n_samples = n_samples;
    m->barcodes = (uint8_t **)calloc(n_samples, sizeof(uint8_t *));
    m->barcode_len = (size_t *)calloc(n_samples, sizeof(size_t));
    m->sample_barcodes = (BitEnc *)calloc(n_samples, sizeof(BitEnc));
    m->max_ns_in_barcodes = 0;
    for (size_t i = 0; i < n_samples; ++i) {
        size_t n = strlen(barcodes[i]);
        m->barcodes[i] = (uint8_t *)malloc(n + 1);
        size_t num_ns = 0;
        for (size_t j = 0; j < n; ++j) {
            m->barcodes[i][j] = to_ascii_uppercase((uint8_t)barcodes[i][j]);
            if (byte_is_nocall(m->barcodes[i][j])) num_ns++;
        }
        m->barcodes[i][n] = 0;
        m->barcode_len[i] = n;
        if (num_ns > m->max_ns_in_barcodes) m->max_ns_in_barcodes = num_ns;
        encode(m->barcodes[i], n, &m->sample_barcodes[i]);
    }
    m->max_mismatches = max_mismatches;
    m->min_mismatch_delta = min_mismatch_delta;
    m->use_cache = use_cache;
    if (use_cache) cache_init(&m->cache, STARTING_CACHE_SIZE);
    return m;
}

void oracle_matcher_free(OracleMatcher *m) {
    if (!m) return;
    for (size_t i = 0; i < m->n_samples; ++i) {
        free(m->barcodes[i]);
        bitenc_free(&m->sample_barcodes[i]);
    }
    free(m->barcodes);
    free(m->barcode_len);
    free(m->sample_barcodes);
    if (m->use_cache) cache_free(&m->cache);
    free(m);
}

size_t oracle_matcher_max_ns(const OracleMatcher *m) { return m->max_ns_in_barcodes; }
uint64_t oracle_matcher_cache_hits(const OracleMatcher *m) { return m->cache_hits; }
uint64_t oracle_matcher_cache_misses(const OracleMatcher *m) { return m->cache_misses; }

/* Return codes shared by assign paths */
#define ORACLE_NONE 0
#define ORACLE_SOME 1
#define ORACLE_ELEN (-1) /* the reference panics: observed length != expected length */

/* barcode_matching.rs:119-160 */
static int assign_internal(const OracleMatcher *m, const uint8_t *read_bases, size_t n, BarcodeMatch *out) {
    uint64_t best_barcode_index = m->n_samples;
    uint8_t best_mismatches = 255;
    uint8_t next_best_mismatches = 255;
    uint8_t max_mismatches = 255;
    BitEnc rb;
    encode(read_bases, n, &rb);
    for (size_t index = 0; index < m->n_samples; ++index) {
        const BitEnc *sample_barcode = &m->sample_barcodes[index];
        /* count_mismatches (:89-110) */
        if (rb.len != sample_barcode->len) {
            bitenc_free(&rb);
            return ORACLE_ELEN;
        }
        uint32_t count = bitenc_hamming(&rb, sample_barcode, (uint32_t)max_mismatches);
        uint8_t mismatches = (uint8_t)count; /* u8::try_from: count <= cap <= 255 always */
        if (mismatches < best_mismatches) {
            next_best_mismatches = best_mismatches;
            best_mismatches = mismatches;
            best_barcode_index = index;
            if (next_best_mismatches < (uint8_t)(255 - m->min_mismatch_delta)) {
                uint8_t cand = (uint8_t)(next_best_mismatches + m->min_mismatch_delta);
                if (cand < max_mismatches) max_mismatches = cand;
            }
        } else if (mismatches < next_best_mismatches) {
            next_best_mismatches = mismatches;
            if (next_best_mismatches < (uint8_t)(255 - m->min_mismatch_delta)) {
                uint8_t cand = (uint8_t)(next_best_mismatches + m->min_mismatch_delta);
                if (cand < max_mismatches) max_mismatches = cand;
            }
        }
    }
    bitenc_free(&rb);
    if (best_mismatches > m->max_mismatches ||
        (uint8_t)(next_best_mismatches - best_mismatches) < m->min_mismatch_delta) {
        return ORACLE_NONE;
    }
    out->best_match = best_barcode_index;
    out->best_mismatches = best_mismatches;
    out->next_best_mismatches = next_best_mismatches;
    return ORACLE_SOME;
}

/* barcode_matching.rs:165-186 */
int oracle_assign(OracleMatcher *m, const uint8_t *read_bases, size_t n, uint64_t *best_match,
                  uint8_t *best_mismatches, uint8_t *next_best_mismatches) {
    if (n < m->barcode_len[0]) return ORACLE_NONE;
    size_t num_no_calls = 0;
    for (size_t i = 0; i < n; ++i) num_no_calls += byte_is_nocall(read_bases[i]) ? 1 : 0;
    if (num_no_calls > (size_t)m->max_mismatches + m->max_ns_in_barcodes) return ORACLE_NONE;
    BarcodeMatch bm;
    int rc;
    if (m->use_cache) {
        uint64_t h = hash_bytes(read_bases, n);
        CacheSlot *s = cache_find(&m->cache, read_bases, (uint32_t)n, h);
        if (s->key) {
            m->cache_hits++;
            bm = s->val;
            rc = ORACLE_SOME;
        } else {
            m->cache_misses++;
            rc = assign_internal(m, read_bases, n, &bm);
            if (rc == ORACLE_SOME) {
                s->key = (uint8_t *)malloc(n ? n : 1);
                memcpy(s->key, read_bases, n);
                s->klen = (uint32_t)n;
                s->hash = h;
                s->val = bm;
                m->cache.used++;
                if (m->cache.used * 2 > m->cache.nslots) cache_grow(&m->cache);
            }
        }
    } else {
        rc = assign_internal(m, read_bases, n, &bm);
    }
    if (rc == ORACLE_SOME) {
        *best_match = bm.best_match;
        *best_mismatches = bm.best_mismatches;
        *next_best_mismatches = bm.next_best_mismatches;
    }
    return rc;
}

/* The demux hot loop (demux.rs:945-977) restated over an SoA buffer of observed barcodes: one
 * assign() per template, Some(m) -> counts[m.best_match] += 1, None -> counts[S] += 1.
 * out_idx[i] = best_match or 0xFFFF for None; out_best/out_next are 255 for None (the reference
 * exposes nothing for None).  `lens` may be NULL (every read has length `stride`).
 * Returns 0, or ORACLE_ELEN at the first read whose length mismatch would panic the reference
 * (index written to *err_index). */
int oracle_assign_batch(OracleMatcher *m, const uint8_t *obs, size_t stride, const uint32_t *lens,
                        uint64_t n, uint16_t *out_idx, uint8_t *out_best, uint8_t *out_next,
                        uint64_t *counts, uint64_t *err_index) {
    for (uint64_t i = 0; i < n; ++i) {
        size_t len = lens ? lens[i] : stride;
        uint64_t bi = 0;
        uint8_t b = 255, nx = 255;
        int rc = oracle_assign(m, obs + i * stride, len, &bi, &b, &nx);
        if (rc == ORACLE_ELEN) {
            if (err_index) *err_index = i;
            return ORACLE_ELEN;
        }
        if (rc == ORACLE_SOME) {
            if (out_idx) out_idx[i] = (uint16_t)bi;
            if (out_best) out_best[i] = b;
            if (out_next) out_next[i] = nx;
            if (counts) counts[bi] += 1;
        } else {
            if (out_idx) out_idx[i] = 0xFFFF;
            if (out_best) out_best[i] = 255;
            if (out_next) out_next[i] = 255;
            if (counts) counts[m->n_samples] += 1;
        }
    }
    return 0;
}
