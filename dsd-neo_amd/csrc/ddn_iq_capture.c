/* ddn_iq_capture.c — reader for dsd-neo's I/Q capture format ("dsd-neo-iq" JSON sidecar + raw cu8 / cf32 / cs16 data),
 * the feeder side of the batched front end (what --iq-replay ingests).
 *
 * reference behaviour followed (src/io/iq/iq_replay.c, docs/iq-capture-replay.md:37-76):
 *   path resolution      a path ending in ".json" is the metadata file, anything else is the data file whose sidecar is
 *                        "<path>.json"; data_file is relative to the sidecar's directory unless absolute (:167-231)
 *   required fields      the 30 keys of metadata_require_required_fields (:1682-1727); unknown keys are ignored
 *   validation           format "dsd-neo-iq", version 1 or 2 (events only with 2), iq_order "IQ", cu8 <-> endianness
 *                        "none", cf32 / cs16 <-> "little" (:1729-1800); sample_rate > 0, base_decimation a power of two
 *                        <= 1024, post_downsample > 0, demod_rate == sample_rate / base_decimation / post_downsample,
 *                        capture_stage one of the two known stages (:675-720)
 *   replayable bytes     min(data_bytes, file size) (file size when data_bytes == 0) rounded down to a whole sample
 *                        (:1859-1882); zero replayable bytes are refused for replay (:1884-1891)
 *   retunes              a capture that contains retunes but carries no event timeline is refused for replay
 * Error codes are the reference's dsd_iq_error values (include/dsd-neo/io/iq_types.h:24-36).
 * Pure host C, no device work: the samples go to ddn_front_end_run* / ddn_cqpsk_run* as they are. */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>

#include "ddn_internal.h"

struct ddn_iq_capture {
    ddn_iq_capture_info info;
    ddn_iq_event* events;
    FILE* fp;
    uint64_t pos;
};

/* ---- JSON tokenizer with the grammar of the reference's (src/io/iq/iq_replay.c:226-523): whitespace is exactly
 * space / tab / LF / CR; strings refuse raw control bytes, take the eight short escapes and \uXXXX for 0x01..0x7f only,
 * and are limited by a 4096-byte buffer; numbers follow RFC 8259 (no leading zeros, digits required after '.', 'e');
 * true / false / null match as prefixes.  The first error is sticky. --------------------------------------------------- */
enum { T_ERR = -1, T_EOF, T_LBRACE, T_RBRACE, T_LBRACKET, T_RBRACKET, T_COMMA, T_COLON, T_STR, T_INT, T_FLOAT, T_TRUE,
       T_FALSE, T_NULL };

typedef struct {
    const char* s;
    size_t n, p;
    int err;
    char buf[4096];
} scan;

typedef struct {
    int type;
    size_t str_len; /* T_STR: bytes in scan.buf (NUL-terminated) */
    const char* num; /* T_INT / T_FLOAT: the literal */
    size_t num_len;
} token;

static int
is_dig(char c) {
    return c >= '0' && c <= '9';
}

static int
tok_string(scan* k, token* t) {
    size_t o = 0;
    k->p++;
    while (k->p < k->n) {
        unsigned char c = (unsigned char)k->s[k->p++];
        if (c < 0x20) {
            return 0;
        }
        if (c == '"') {
            k->buf[o] = 0;
            t->type = T_STR;
            t->str_len = o;
            return 1;
        }
        if (c == '\\') {
            if (k->p >= k->n) {
                return 0;
            }
            const char e = k->s[k->p++];
            switch (e) {
                case '"': c = '"'; break;
                case '\\': c = '\\'; break;
                case '/': c = '/'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'n': c = '\n'; break;
                case 'r': c = '\r'; break;
                case 't': c = '\t'; break;
                case 'u': {
                    if (k->p + 4 > k->n) {
                        return 0;
                    }
                    unsigned v = 0;
                    for (int i = 0; i < 4; i++) {
                        const char h = k->s[k->p + i];
                        if (!isxdigit((unsigned char)h)) {
                            return 0;
                        }
                        v = v * 16 + (unsigned)(is_dig(h) ? h - '0' : (tolower((unsigned char)h) - 'a' + 10));
                    }
                    if (v == 0 || v > 0x7F) {
                        return 0;
                    }
                    k->p += 4;
                    c = (unsigned char)v;
                    break;
                }
                default: return 0;
            }
        }
        if (o + 1 >= sizeof(k->buf)) {
            return 0;
        }
        k->buf[o++] = (char)c;
    }
    return 0; /* unterminated */
}

static int
tok_number(scan* k, token* t) {
    const size_t start = k->p;
    size_t p = k->p;
    int is_float = 0;
    if (k->s[p] == '-') {
        p++;
        if (p >= k->n || !is_dig(k->s[p])) {
            return 0;
        }
    }
    if (p < k->n && k->s[p] == '0') {
        p++;
    } else {
        if (p >= k->n || !is_dig(k->s[p])) {
            return 0;
        }
        while (p < k->n && is_dig(k->s[p])) {
            p++;
        }
    }
    if (p < k->n && k->s[p] == '.') {
        is_float = 1;
        p++;
        if (p >= k->n || !is_dig(k->s[p])) {
            return 0;
        }
        while (p < k->n && is_dig(k->s[p])) {
            p++;
        }
    }
    if (p < k->n && (k->s[p] == 'e' || k->s[p] == 'E')) {
        is_float = 1;
        p++;
        if (p < k->n && (k->s[p] == '+' || k->s[p] == '-')) {
            p++;
        }
        if (p >= k->n || !is_dig(k->s[p])) {
            return 0;
        }
        while (p < k->n && is_dig(k->s[p])) {
            p++;
        }
    }
    t->type = is_float ? T_FLOAT : T_INT;
    t->num = k->s + start;
    t->num_len = p - start;
    k->p = p;
    return 1;
}

static token
next_tok(scan* k) {
    token t;
    memset(&t, 0, sizeof(t));
    t.type = T_ERR;
    if (k->err) {
        return t;
    }
    while (k->p < k->n && (k->s[k->p] == ' ' || k->s[k->p] == '\t' || k->s[k->p] == '\n' || k->s[k->p] == '\r')) {
        k->p++;
    }
    if (k->p >= k->n) {
        t.type = T_EOF;
        return t;
    }
    const char c = k->s[k->p];
    int simple = T_ERR;
    switch (c) {
        case '{': simple = T_LBRACE; break;
        case '}': simple = T_RBRACE; break;
        case '[': simple = T_LBRACKET; break;
        case ']': simple = T_RBRACKET; break;
        case ',': simple = T_COMMA; break;
        case ':': simple = T_COLON; break;
        default: break;
    }
    if (simple != T_ERR) {
        k->p++;
        t.type = simple;
        return t;
    }
    int ok = 0;
    if (c == '"') {
        ok = tok_string(k, &t);
    } else if (c == '-' || is_dig(c)) {
        ok = tok_number(k, &t);
    } else if (k->p + 4 <= k->n && memcmp(k->s + k->p, "true", 4) == 0) {
        k->p += 4;
        t.type = T_TRUE;
        ok = 1;
    } else if (k->p + 5 <= k->n && memcmp(k->s + k->p, "false", 5) == 0) {
        k->p += 5;
        t.type = T_FALSE;
        ok = 1;
    } else if (k->p + 4 <= k->n && memcmp(k->s + k->p, "null", 4) == 0) {
        k->p += 4;
        t.type = T_NULL;
        ok = 1;
    }
    if (!ok) {
        k->err = 1;
        t.type = T_ERR;
    }
    return t;
}

/* string value into a field of `cap` bytes (copy_token_to_buffer, iq_replay.c:530-553): too long or, for every field
 * but "notes", a control byte that came in through an escape -> invalid metadata */
static int
tok_to_str(const scan* k, const token* t, char* out, size_t cap, int reject_controls) {
    if (t->type != T_STR || t->str_len + 1 > cap) {
        return -1;
    }
    if (reject_controls) {
        for (size_t i = 0; i < t->str_len; i++) {
            if ((unsigned char)k->buf[i] < 0x20) {
                return -1;
            }
        }
    }
    memcpy(out, k->buf, t->str_len);
    out[t->str_len] = 0;
    return 0;
}

/* token_to_u64 / token_to_u32 / token_to_i32 (iq_replay.c:555-621): integer tokens of at most 63 characters through
 * strtoull / strtol */
static int
tok_to_u64(const token* t, uint64_t* out, uint64_t max) {
    char b[64];
    if (t->type != T_INT || t->num_len == 0 || t->num_len >= sizeof(b) || t->num[0] == '-') {
        return -1;
    }
    memcpy(b, t->num, t->num_len);
    b[t->num_len] = 0;
    errno = 0;
    char* end = NULL;
    const unsigned long long v = strtoull(b, &end, 10);
    if (errno != 0 || !end || *end || v > max) {
        return -1;
    }
    *out = (uint64_t)v;
    return 0;
}

static int
tok_to_i32(const token* t, int* out) {
    char b[64];
    if (t->type != T_INT || t->num_len == 0 || t->num_len >= sizeof(b)) {
        return -1;
    }
    memcpy(b, t->num, t->num_len);
    b[t->num_len] = 0;
    errno = 0;
    char* end = NULL;
    const long v = strtol(b, &end, 10);
    if (errno != 0 || !end || *end || v < INT_MIN || v > INT_MAX) {
        return -1;
    }
    *out = (int)v;
    return 0;
}

static int
tok_to_bool(const token* t, int* out) {
    if (t->type != T_TRUE && t->type != T_FALSE) {
        return -1;
    }
    *out = t->type == T_TRUE;
    return 0;
}

/* "key" ':' value-token of an object, or its closing brace (a trailing comma before '}' is accepted, as in
 * metadata_parse_key_value / parse_event_key_value).  Returns 1 = pair read, 0 = object closed, -1 = malformed. */
static int
next_pair(scan* k, char* key, size_t key_cap, token* val) {
    const token kt = next_tok(k);
    if (k->err) {
        return -1;
    }
    if (kt.type == T_RBRACE) {
        return 0;
    }
    if (kt.type != T_STR || kt.str_len + 1 > key_cap) {
        return -1;
    }
    memcpy(key, k->buf, kt.str_len + 1);
    if (next_tok(k).type != T_COLON) {
        return -1;
    }
    *val = next_tok(k);
    return k->err ? -1 : 1;
}

#define FAIL(code, ...)                                                                                                \
    do {                                                                                                               \
        ddn_set_error(__VA_ARGS__);                                                                                    \
        return (code);                                                                                                 \
    } while (0)

static size_t
align_of(int fmt) {
    return fmt == DDN_IQ_FORMAT_CU8 ? 2 : (fmt == DDN_IQ_FORMAT_CF32 ? 8 : (fmt == DDN_IQ_FORMAT_CS16 ? 4 : 0));
}

int
ddn_iq_effective_bytes(uint64_t data_bytes, uint64_t actual_file_size, int sample_format, uint64_t* out_effective,
                       int* out_size_mismatch) {
    const size_t a = align_of(sample_format);
    if (!out_effective || a == 0) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    uint64_t raw = actual_file_size;
    int mismatch = 0;
    if (data_bytes > 0) {
        raw = data_bytes < actual_file_size ? data_bytes : actual_file_size;
        mismatch = data_bytes != actual_file_size;
    }
    *out_effective = raw - (raw % (uint64_t)a);
    if (out_size_mismatch) {
        *out_size_mismatch = mismatch;
    }
    return DDN_IQ_OK;
}

/* the "events" array after its '[' (parse_events_array / parse_event_object, iq_replay.c:958-1058) */
static int
parse_events(scan* k, ddn_iq_event** out, uint32_t* count) {
    ddn_iq_event* ev = NULL;
    uint32_t n = 0, cap = 0;
    token t = next_tok(k);
    if (t.type == T_RBRACKET) {
        *out = NULL;
        *count = 0;
        return DDN_IQ_OK;
    }
    for (;;) {
        if (t.type != T_LBRACE) {
            free(ev);
            FAIL(DDN_IQ_ERR_INVALID_META, "event %u is not an object", n);
        }
        ddn_iq_event e;
        memset(&e, 0, sizeof(e));
        unsigned seen = 0; /* 1 kind, 2 byte_offset, 4 duration, 8 center, 16 capture center, 32 rate, 64 reason */
        for (;;) {
            char key[128];
            token v;
            const int pr = next_pair(k, key, sizeof(key), &v);
            if (pr < 0 || (pr > 0 && (v.type == T_LBRACE || v.type == T_LBRACKET))) {
                free(ev);
                FAIL(DDN_IQ_ERR_INVALID_META, "malformed field in event %u", n);
            }
            if (pr == 0) {
                break;
            }
            uint64_t u = 0;
            int rc = 0;
            if (!strcmp(key, "kind")) {
                char kind[32];
                rc = tok_to_str(k, &v, kind, sizeof(kind), 1);
                if (rc == 0) {
                    e.kind = !strcmp(kind, "RETUNE") ? DDN_IQ_EVENT_RETUNE
                                                     : (!strcmp(kind, "MUTE") ? DDN_IQ_EVENT_MUTE
                                                                              : (!strcmp(kind, "RESET") ? DDN_IQ_EVENT_RESET : 0));
                    rc = e.kind == 0 ? -1 : 0;
                }
                seen |= 1;
            } else if (!strcmp(key, "reason")) {
                rc = tok_to_str(k, &v, e.reason, sizeof(e.reason), 1);
                seen |= 64;
            } else if (!strcmp(key, "byte_offset")) {
                rc = tok_to_u64(&v, &u, UINT64_MAX);
                e.byte_offset = u;
                seen |= 2;
            } else if (!strcmp(key, "duration_bytes")) {
                rc = tok_to_u64(&v, &u, UINT64_MAX);
                e.duration_bytes = u;
                seen |= 4;
            } else if (!strcmp(key, "center_frequency_hz")) {
                rc = tok_to_u64(&v, &u, UINT64_MAX);
                e.center_frequency_hz = u;
                seen |= 8;
            } else if (!strcmp(key, "capture_center_frequency_hz")) {
                rc = tok_to_u64(&v, &u, UINT64_MAX);
                e.capture_center_frequency_hz = u;
                seen |= 16;
            } else if (!strcmp(key, "sample_rate_hz")) {
                rc = tok_to_u64(&v, &u, 0xFFFFFFFFull);
                e.sample_rate_hz = (uint32_t)u;
                seen |= 32;
            } /* any other key: whatever single token followed the colon is ignored, as in the reference */
            if (rc != 0) {
                free(ev);
                FAIL(DDN_IQ_ERR_INVALID_META, "bad value for '%s' in event %u", key, n);
            }
            const token d = next_tok(k);
            if (d.type == T_RBRACE) {
                break;
            }
            if (d.type != T_COMMA) {
                free(ev);
                FAIL(DDN_IQ_ERR_INVALID_META, "bad delimiter in event %u", n);
            }
        }
        /* validate_event_object_fields, src/io/iq/iq_replay.c:935-958 */
        if ((seen & 67) != 67 || (e.kind == DDN_IQ_EVENT_MUTE && (!(seen & 4) || e.duration_bytes == 0))
            || (e.kind != DDN_IQ_EVENT_MUTE
                && ((seen & 56) != 56 || e.center_frequency_hz == 0 || e.capture_center_frequency_hz == 0
                    || e.sample_rate_hz == 0))) {
            free(ev);
            FAIL(DDN_IQ_ERR_INVALID_META, "event %u lacks a required field", n);
        }
        if (n == cap) {
            cap = cap ? cap * 2 : 8;
            ddn_iq_event* t2 = (ddn_iq_event*)realloc(ev, sizeof(*t2) * cap);
            if (!t2) {
                free(ev);
                FAIL(DDN_IQ_ERR_ALLOC, "out of memory");
            }
            ev = t2;
        }
        ev[n++] = e;
        t = next_tok(k);
        if (t.type == T_COMMA) {
            t = next_tok(k);
            continue;
        }
        if (t.type == T_RBRACKET) {
            break;
        }
        free(ev);
        FAIL(DDN_IQ_ERR_INVALID_META, "bad delimiter in events array");
    }
    *out = ev;
    *count = n;
    return DDN_IQ_OK;
}

static int
parse_metadata(const char* text, size_t len, ddn_iq_capture_info* c, ddn_iq_event** events, char* data_file,
               size_t data_file_cap) {
    scan* kp = (scan*)calloc(1, sizeof(scan));
    if (!kp) {
        FAIL(DDN_IQ_ERR_ALLOC, "out of memory");
    }
    kp->s = text;
    kp->n = len;
    char format[64] = "", sfmt[32] = "", order[16] = "", endian[16] = "";
    char backend[32], args[256], started[64], notes[256];
    int version = 0, combine_rotate = 1, have_events = 0;
    unsigned long long seen = 0;
    static const char* const req[] = {"format", "version", "sample_format", "iq_order", "endianness", "capture_stage",
                                      "sample_rate_hz", "center_frequency_hz", "capture_center_frequency_hz", "ppm",
                                      "tuner_gain_tenth_db", "rtl_dsp_bw_khz", "base_decimation", "post_downsample",
                                      "demod_rate_hz", "offset_tuning_enabled", "fs4_shift_enabled",
                                      "combine_rotate_enabled", "muted_bytes_excluded", "contains_retunes",
                                      "capture_retune_count", "source_backend", "source_args", "capture_started_utc",
                                      "data_file", "data_bytes", "capture_drops", "capture_drop_blocks",
                                      "input_ring_drops", "notes"};
    const int n_req = (int)(sizeof(req) / sizeof(req[0]));
#define BAIL(code, ...)                                                                                                \
    do {                                                                                                               \
        free(kp);                                                                                                      \
        FAIL(code, __VA_ARGS__);                                                                                       \
    } while (0)
    if (next_tok(kp).type != T_LBRACE) {
        BAIL(DDN_IQ_ERR_INVALID_META, "metadata is not a JSON object");
    }
    for (;;) {
        char key[128];
        token v;
        const int pr = next_pair(kp, key, sizeof(key), &v);
        if (pr < 0) {
            BAIL(DDN_IQ_ERR_INVALID_META, "malformed key / value near offset %zu", kp->p);
        }
        if (pr == 0) {
            break;
        }
        for (int i = 0; i < n_req; i++) {
            if (!strcmp(key, req[i])) {
                seen |= 1ull << i;
            }
        }
        uint64_t u = 0;
        int rc = 0, b = 0;
#define STR(field, cap) (rc = tok_to_str(kp, &v, field, cap, 1))
#define U64(dst)                                                                                                       \
    do {                                                                                                               \
        rc = tok_to_u64(&v, &u, UINT64_MAX);                                                                           \
        (dst) = u;                                                                                                     \
    } while (0)
#define U32(dst)                                                                                                       \
    do {                                                                                                               \
        rc = tok_to_u64(&v, &u, 0xFFFFFFFFull);                                                                        \
        (dst) = (uint32_t)u;                                                                                           \
    } while (0)
#define I32(dst) (rc = tok_to_i32(&v, &(dst)))
#define BOOL(dst)                                                                                                      \
    do {                                                                                                               \
        rc = tok_to_bool(&v, &b);                                                                                      \
        (dst) = b;                                                                                                     \
    } while (0)
        if (!strcmp(key, "format")) {
            STR(format, sizeof(format));
        } else if (!strcmp(key, "version")) {
            I32(version);
        } else if (!strcmp(key, "sample_format")) {
            STR(sfmt, sizeof(sfmt));
        } else if (!strcmp(key, "iq_order")) {
            STR(order, sizeof(order));
        } else if (!strcmp(key, "endianness")) {
            STR(endian, sizeof(endian));
        } else if (!strcmp(key, "capture_stage")) {
            STR(c->capture_stage, sizeof(c->capture_stage));
        } else if (!strcmp(key, "sample_rate_hz")) {
            U32(c->sample_rate_hz);
        } else if (!strcmp(key, "center_frequency_hz")) {
            U64(c->center_frequency_hz);
        } else if (!strcmp(key, "capture_center_frequency_hz")) {
            U64(c->capture_center_frequency_hz);
        } else if (!strcmp(key, "ppm")) {
            I32(c->ppm);
        } else if (!strcmp(key, "tuner_gain_tenth_db")) {
            I32(c->tuner_gain_tenth_db);
        } else if (!strcmp(key, "rtl_dsp_bw_khz")) {
            I32(c->rtl_dsp_bw_khz);
        } else if (!strcmp(key, "base_decimation")) {
            U32(c->base_decimation);
        } else if (!strcmp(key, "post_downsample")) {
            U32(c->post_downsample);
        } else if (!strcmp(key, "demod_rate_hz")) {
            U32(c->demod_rate_hz);
        } else if (!strcmp(key, "offset_tuning_enabled")) {
            BOOL(c->offset_tuning_enabled);
        } else if (!strcmp(key, "fs4_shift_enabled")) {
            BOOL(c->fs4_shift_enabled);
        } else if (!strcmp(key, "combine_rotate_enabled")) {
            BOOL(combine_rotate);
        } else if (!strcmp(key, "muted_bytes_excluded")) {
            BOOL(c->muted_bytes_excluded);
        } else if (!strcmp(key, "contains_retunes")) {
            BOOL(c->contains_retunes);
        } else if (!strcmp(key, "size_limit_reached")) {
            BOOL(c->size_limit_reached);
        } else if (!strcmp(key, "capture_retune_count")) {
            U32(c->capture_retune_count);
        } else if (!strcmp(key, "data_file")) {
            STR(data_file, data_file_cap < 2048 ? data_file_cap : 2048);
        } else if (!strcmp(key, "data_bytes")) {
            U64(c->data_bytes);
        } else if (!strcmp(key, "capture_drops")) {
            U64(c->capture_drops);
        } else if (!strcmp(key, "capture_drop_blocks")) {
            U64(c->capture_drop_blocks);
        } else if (!strcmp(key, "input_ring_drops")) {
            U64(c->input_ring_drops);
        } else if (!strcmp(key, "source_backend")) { /* these four must be strings that fit the reference's fields */
            STR(backend, sizeof(backend));
        } else if (!strcmp(key, "source_args")) {
            STR(args, sizeof(args));
        } else if (!strcmp(key, "capture_started_utc")) {
            STR(started, sizeof(started));
        } else if (!strcmp(key, "notes")) {
            rc = v.type == T_NULL ? 0 : tok_to_str(kp, &v, notes, sizeof(notes), 0);
        } else if (!strcmp(key, "events")) {
            if (v.type != T_LBRACKET) {
                BAIL(DDN_IQ_ERR_INVALID_META, "field 'events' expects array");
            }
            free(*events);
            *events = NULL;
            c->event_count = 0;
            const int er = parse_events(kp, events, &c->event_count);
            if (er != DDN_IQ_OK) {
                free(kp);
                return er;
            }
            have_events = 1;
        } else if (v.type == T_LBRACE || v.type == T_LBRACKET) {
            /* unknown keys: any single token is ignored; nested values are refused like the reference does */
            BAIL(DDN_IQ_ERR_INVALID_META, "nested structures are unsupported in metadata (key '%s')", key);
        }
#undef STR
#undef U64
#undef U32
#undef I32
#undef BOOL
        if (rc != 0) {
            BAIL(DDN_IQ_ERR_INVALID_META, "bad value for '%s'", key);
        }
        const token d = next_tok(kp);
        if (d.type == T_RBRACE) {
            break;
        }
        if (d.type != T_COMMA) {
            BAIL(DDN_IQ_ERR_INVALID_META, "bad delimiter after '%s'", key);
        }
    }
    if (next_tok(kp).type != T_EOF) {
        BAIL(DDN_IQ_ERR_INVALID_META, "trailing JSON content at offset %zu", kp->p);
    }
    free(kp);
#undef BAIL
    for (int i = 0; i < n_req; i++) {
        if (!(seen & (1ull << i))) {
            FAIL(DDN_IQ_ERR_INVALID_META, "missing required field '%s'", req[i]);
        }
    }
    if (strcmp(format, "dsd-neo-iq") != 0) {
        FAIL(DDN_IQ_ERR_INVALID_META, "unsupported format '%s'", format);
    }
    if (version != 1 && version != 2) {
        FAIL(DDN_IQ_ERR_UNSUPPORTED_VER, "unsupported metadata version %d", version);
    }
    c->metadata_version = (uint32_t)version;
    if (have_events && version != 2) {
        FAIL(DDN_IQ_ERR_UNSUPPORTED_VER, "events require metadata version 2");
    }
    if (strcmp(order, "IQ") != 0) {
        FAIL(DDN_IQ_ERR_INVALID_META, "unsupported iq_order '%s'", order);
    }
    const int is_cu8 = !strcmp(sfmt, "cu8"), is_cf32 = !strcmp(sfmt, "cf32"), is_cs16 = !strcmp(sfmt, "cs16");
    if (!is_cu8 && !is_cf32 && !is_cs16) {
        FAIL(DDN_IQ_ERR_UNSUPPORTED_FMT, "unsupported sample_format '%s'", sfmt);
    }
    c->sample_format = is_cu8 ? DDN_IQ_FORMAT_CU8 : (is_cf32 ? DDN_IQ_FORMAT_CF32 : DDN_IQ_FORMAT_CS16);
    if (strcmp(endian, is_cu8 ? "none" : "little") != 0) {
        FAIL(DDN_IQ_ERR_INVALID_META, "%s requires endianness '%s'", sfmt, is_cu8 ? "none" : "little");
    }
    c->combine_rotate_enabled = combine_rotate;
    /* validate_replay_semantics */
    if (c->sample_rate_hz == 0) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "sample_rate_hz must be > 0");
    }
    if (c->base_decimation == 0 || (c->base_decimation & (c->base_decimation - 1u)) != 0) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "base_decimation must be a power of two");
    }
    if (c->base_decimation > 1024u) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "base_decimation (%u) exceeds the maximum of 1024", c->base_decimation);
    }
    if (c->post_downsample == 0) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "post_downsample must be > 0");
    }
    if (c->demod_rate_hz == 0) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "demod_rate_hz must be > 0");
    }
    if ((uint64_t)c->sample_rate_hz / c->base_decimation / c->post_downsample != c->demod_rate_hz) {
        FAIL(DDN_IQ_ERR_RATE_CHAIN, "demod_rate_hz (%u) inconsistent with sample_rate/base_decimation/post_downsample",
             c->demod_rate_hz);
    }
    if (strcmp(c->capture_stage, "post_mute_pre_widen") != 0 && strcmp(c->capture_stage, "post_driver_cf32_pre_ring") != 0) {
        FAIL(DDN_IQ_ERR_UNSUPPORTED_FMT, "unsupported capture_stage '%s'", c->capture_stage);
    }
    return DDN_IQ_OK;
}

static int
resolve_paths(const char* path, ddn_iq_capture_info* c) {
    const size_t n = strlen(path);
    if (n >= 5 && strcmp(path + n - 5, ".json") == 0) {
        if (n + 1 > sizeof(c->metadata_path)) {
            FAIL(DDN_IQ_ERR_INVALID_ARG, "metadata path too long");
        }
        memcpy(c->metadata_path, path, n + 1);
    } else {
        if (n + 6 > sizeof(c->metadata_path)) {
            FAIL(DDN_IQ_ERR_INVALID_ARG, "metadata path too long");
        }
        snprintf(c->metadata_path, sizeof(c->metadata_path), "%s.json", path);
        struct stat st;
        if (stat(c->metadata_path, &st) != 0) {
            FAIL(DDN_IQ_ERR_IO, "metadata sidecar not found for '%s' (expected '%s')", path, c->metadata_path);
        }
    }
    return DDN_IQ_OK;
}

/* validate_replay_events_metadata, src/io/iq/iq_replay.c:1065-1232: sorted, inside the replayable bytes, sample
 * aligned; MUTE needs a positive aligned duration; RETUNE / RESET need frequencies and the capture's sample rate; for
 * replay every RETUNE must be followed by a RESET and their number must equal capture_retune_count */
static int
validate_events(const ddn_iq_capture_info* c, const ddn_iq_event* ev, uint64_t max_offset, int check_max, int reject) {
    if (c->event_count > 0 && c->metadata_version != 2) {
        FAIL(DDN_IQ_ERR_UNSUPPORTED_VER, "events require metadata version 2");
    }
    if (reject && (c->contains_retunes || c->capture_retune_count > 0) && c->event_count == 0) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture contains retunes but has no replay event timeline");
    }
    const size_t a = align_of(c->sample_format);
    uint64_t prev = 0;
    int has_freq = 0, needs_reset = 0;
    uint32_t retunes = 0, completed = 0;
    for (uint32_t i = 0; i < c->event_count; i++) {
        const ddn_iq_event* e = &ev[i];
        if (i > 0 && e->byte_offset < prev) {
            FAIL(DDN_IQ_ERR_INVALID_META, "IQ events are not sorted by byte_offset");
        }
        prev = e->byte_offset;
        if (check_max && e->byte_offset > max_offset) {
            FAIL(DDN_IQ_ERR_INVALID_META, "IQ event byte_offset exceeds replay bytes");
        }
        if (e->byte_offset % a) {
            FAIL(DDN_IQ_ERR_ALIGNMENT, "IQ event byte_offset is not aligned to sample format");
        }
        if (e->kind == DDN_IQ_EVENT_MUTE) {
            if (e->duration_bytes == 0) {
                FAIL(DDN_IQ_ERR_INVALID_META, "MUTE event duration_bytes must be > 0");
            }
            if (e->duration_bytes % a) {
                FAIL(DDN_IQ_ERR_ALIGNMENT, "MUTE event duration_bytes is not aligned to sample format");
            }
            continue;
        }
        has_freq = 1;
        if (e->center_frequency_hz == 0 || e->capture_center_frequency_hz == 0 || e->sample_rate_hz == 0) {
            FAIL(DDN_IQ_ERR_INVALID_META, "RETUNE/RESET event is missing frequency or sample-rate fields");
        }
        if (e->sample_rate_hz != c->sample_rate_hz) {
            FAIL(DDN_IQ_ERR_RATE_CHAIN, "event sample_rate_hz changes are not supported for replay");
        }
        if (e->kind == DDN_IQ_EVENT_RETUNE) {
            if (reject && needs_reset) {
                FAIL(DDN_IQ_ERR_RETUNE_REJECT, "RETUNE event is missing a following RESET event");
            }
            retunes++;
            needs_reset = 1;
        } else if (needs_reset) {
            completed++;
            needs_reset = 0;
        }
    }
    if (reject && c->contains_retunes && !has_freq) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture contains retunes but event timeline has no RETUNE/RESET event");
    }
    if (!reject || (!c->contains_retunes && c->capture_retune_count == 0 && retunes == 0)) {
        return DDN_IQ_OK;
    }
    if (needs_reset) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture retune timeline is missing a RESET event");
    }
    if (c->capture_retune_count == 0) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture contains retunes but capture_retune_count is zero");
    }
    if (retunes != c->capture_retune_count) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture retune count does not match RETUNE events");
    }
    if (completed != c->capture_retune_count) {
        FAIL(DDN_IQ_ERR_RETUNE_REJECT, "capture retune timeline is missing a RESET event");
    }
    return DDN_IQ_OK;
}

static int
load_info(const char* path, ddn_iq_capture_info* c, ddn_iq_event** events, int reject_missing_timeline) {
    memset(c, 0, sizeof(*c));
    *events = NULL;
    if (!path) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    int rc = resolve_paths(path, c);
    if (rc != DDN_IQ_OK) {
        return rc;
    }
    FILE* f = fopen(c->metadata_path, "rb");
    if (!f) {
        FAIL(DDN_IQ_ERR_IO, "cannot open metadata '%s': %s", c->metadata_path, strerror(errno));
    }
    fseeko(f, 0, SEEK_END);
    const off_t sz = ftello(f);
    fseeko(f, 0, SEEK_SET);
    if (sz <= 0 || sz > (off_t)(16 << 20)) {
        fclose(f);
        FAIL(DDN_IQ_ERR_INVALID_META, "metadata file is empty or implausibly large");
    }
    char* text = (char*)malloc((size_t)sz + 1);
    if (!text || fread(text, 1, (size_t)sz, f) != (size_t)sz) {
        free(text);
        fclose(f);
        FAIL(DDN_IQ_ERR_IO, "cannot read metadata '%s'", c->metadata_path);
    }
    fclose(f);
    text[sz] = 0;
    char data_file[1024] = "";
    rc = parse_metadata(text, (size_t)sz, c, events, data_file, sizeof(data_file));
    free(text);
    if (rc != DDN_IQ_OK) {
        free(*events);
        *events = NULL;
        return rc;
    }
    /* data_file is relative to the sidecar's directory unless absolute; "absolute" and the directory separator take the
     * Windows spellings too, on every host (path_is_absolute / resolve_data_path, src/io/iq/iq_replay.c:66-78,194-224) */
    const char* slash = strrchr(c->metadata_path, '/');
    const char* bslash = strrchr(c->metadata_path, '\\');
    if (bslash && (!slash || bslash > slash)) {
        slash = bslash;
    }
    const int absolute = data_file[0] == '/' || data_file[0] == '\\'
                         || (isalpha((unsigned char)data_file[0]) && (unsigned char)data_file[0] < 0x80 && data_file[1] == ':');
    if (absolute || !slash) {
        snprintf(c->data_path, sizeof(c->data_path), "%s", data_file);
    } else {
        const size_t dir = (size_t)(slash - c->metadata_path + 1);
        if (dir + strlen(data_file) + 1 > sizeof(c->data_path)) {
            free(*events);
            *events = NULL;
            FAIL(DDN_IQ_ERR_INVALID_ARG, "resolved data path too long");
        }
        memcpy(c->data_path, c->metadata_path, dir);
        strcpy(c->data_path + dir, data_file);
    }
    rc = validate_events(c, *events, c->data_bytes, c->data_bytes > 0, reject_missing_timeline);
    if (rc != DDN_IQ_OK) {
        free(*events);
        *events = NULL;
        return rc;
    }
    struct stat st;
    c->actual_file_bytes = (stat(c->data_path, &st) == 0 && S_ISREG(st.st_mode)) ? (uint64_t)st.st_size : 0;
    ddn_iq_effective_bytes(c->data_bytes, c->actual_file_bytes, c->sample_format, &c->effective_bytes, &c->size_mismatch);
    return DDN_IQ_OK;
}

int
ddn_iq_capture_read_info(const char* path, ddn_iq_capture_info* out_info) {
    if (!out_info) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    ddn_iq_event* ev = NULL;
    const int rc = load_info(path, out_info, &ev, 0);
    free(ev);
    return rc;
}

int
ddn_iq_capture_open(const char* path, ddn_iq_capture** out) {
    if (!out) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    *out = NULL;
    ddn_iq_capture* c = (ddn_iq_capture*)calloc(1, sizeof(*c));
    if (!c) {
        return DDN_IQ_ERR_ALLOC;
    }
    int rc = load_info(path, &c->info, &c->events, 1);
    if (rc == DDN_IQ_OK && c->info.actual_file_bytes == 0) {
        struct stat st;
        if (stat(c->info.data_path, &st) != 0 || !S_ISREG(st.st_mode)) {
            ddn_set_error("cannot stat data file '%s'", c->info.data_path);
            rc = DDN_IQ_ERR_IO;
        }
    }
    if (rc == DDN_IQ_OK && c->info.effective_bytes == 0) {
        ddn_set_error("no replayable bytes in '%s'", c->info.data_path);
        rc = DDN_IQ_ERR_ALIGNMENT;
    }
    if (rc == DDN_IQ_OK) {
        rc = validate_events(&c->info, c->events, c->info.effective_bytes, 1, 1);
    }
    if (rc == DDN_IQ_OK) {
        c->fp = fopen(c->info.data_path, "rb");
        if (!c->fp) {
            ddn_set_error("cannot open data file '%s': %s", c->info.data_path, strerror(errno));
            rc = DDN_IQ_ERR_IO;
        }
    }
    if (rc != DDN_IQ_OK) {
        free(c->events);
        free(c);
        return rc;
    }
    *out = c;
    return DDN_IQ_OK;
}

void
ddn_iq_capture_close(ddn_iq_capture* c) {
    if (!c) {
        return;
    }
    if (c->fp) {
        fclose(c->fp);
    }
    free(c->events);
    free(c);
}

const ddn_iq_capture_info*
ddn_iq_capture_get_info(const ddn_iq_capture* c) {
    return c ? &c->info : NULL;
}

const ddn_iq_event*
ddn_iq_capture_get_events(const ddn_iq_capture* c, uint32_t* out_count) {
    if (out_count) {
        *out_count = c ? c->info.event_count : 0;
    }
    return c ? c->events : NULL;
}

int
ddn_iq_capture_read(ddn_iq_capture* c, void* out, size_t max_bytes, size_t* out_bytes) {
    if (out_bytes) {
        *out_bytes = 0;
    }
    if (!c || !out || !out_bytes) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    /* like dsd_iq_replay_read: any byte count, capped at what is left of the replayable bytes; 0 bytes = end */
    const uint64_t left = c->info.effective_bytes - c->pos;
    size_t want = max_bytes;
    if ((uint64_t)want > left) {
        want = (size_t)left;
    }
    if (want == 0) {
        return DDN_IQ_OK;
    }
    const size_t got = fread(out, 1, want, c->fp);
    if (got == 0 && ferror(c->fp)) {
        FAIL(DDN_IQ_ERR_IO, "read error on '%s'", c->info.data_path);
    }
    c->pos += got;
    *out_bytes = got;
    return DDN_IQ_OK;
}

int
ddn_iq_capture_rewind(ddn_iq_capture* c) {
    if (!c || fseeko(c->fp, 0, SEEK_SET) != 0) {
        return DDN_IQ_ERR_IO;
    }
    c->pos = 0;
    return DDN_IQ_OK;
}

/* Capture-side FS/4 shift of a CU8 row: pair n times j^n from the start of the capture, in the byte domain (negation is
 * 255 - b, exact under the -127.5 bias) - what the reference's replay does before the widen, or fused into it
 * (src/io/radio/rtl_device.cpp:479-500 policy, :518-560 two-pass rotation; src/dsp/simd_widen.cpp:167-199 combined).  The
 * two give identical floats, so one byte pass covers both settings of combine_rotate_enabled. */
static void
fs4_rotate_cu8(unsigned char* row, size_t pairs) {
    for (size_t n = 0; n < pairs; n++) {
        const unsigned char i = row[2 * n], q = row[2 * n + 1];
        switch (n & 3u) {
            case 0: break;
            case 1:
                row[2 * n] = (unsigned char)(255u - q);
                row[2 * n + 1] = i;
                break;
            case 2:
                row[2 * n] = (unsigned char)(255u - i);
                row[2 * n + 1] = (unsigned char)(255u - q);
                break;
            default:
                row[2 * n] = q;
                row[2 * n + 1] = (unsigned char)(255u - i);
                break;
        }
    }
}

/* B captures with one sample format and one rate chain -> one channel-major buffer [B][n] (n = the shortest capture's
 * complex sample count), the shape ddn_front_end_run_host / ddn_cqpsk_run_host take.  *out_buf is malloc'ed.
 * The metadata's capture-side transforms are honoured, not dropped: fs4_shift_enabled on a CU8 capture rotates each row
 * here (the batch kernels then see what the reference's demod thread sees); base_decimation is the caller's to configure
 * on the front end (returned in out_info0); a post_downsample other than 1 has no stage in this library and is refused. */
int
ddn_iq_load_batch(const char* const* paths, int n_captures, void** out_buf, size_t* out_n_samples,
                  ddn_iq_capture_info* out_info0) {
    if (!paths || n_captures <= 0 || !out_buf || !out_n_samples) {
        return DDN_IQ_ERR_INVALID_ARG;
    }
    *out_buf = NULL;
    *out_n_samples = 0;
    ddn_iq_capture** caps = (ddn_iq_capture**)calloc((size_t)n_captures, sizeof(*caps));
    if (!caps) {
        return DDN_IQ_ERR_ALLOC;
    }
    int rc = DDN_IQ_OK;
    uint64_t n_min = UINT64_MAX;
    for (int i = 0; i < n_captures && rc == DDN_IQ_OK; i++) {
        rc = ddn_iq_capture_open(paths[i], &caps[i]);
        if (rc != DDN_IQ_OK) {
            break;
        }
        const ddn_iq_capture_info* a = &caps[i]->info;
        const ddn_iq_capture_info* b = &caps[0]->info;
        if (a->post_downsample != 1) {
            ddn_set_error("capture %d ('%s'): post_downsample = %u has no stage in the batch front end (only 1 is supported)", i,
                          paths[i], a->post_downsample);
            rc = DDN_IQ_ERR_RATE_CHAIN;
            break;
        }
        if (a->sample_format != b->sample_format || a->sample_rate_hz != b->sample_rate_hz
            || a->base_decimation != b->base_decimation || a->fs4_shift_enabled != b->fs4_shift_enabled) {
            ddn_set_error("capture %d ('%s') has a different sample format or rate chain than capture 0", i, paths[i]);
            rc = DDN_IQ_ERR_RATE_CHAIN;
            break;
        }
        const uint64_t n = a->effective_bytes / align_of(a->sample_format);
        n_min = n < n_min ? n : n_min;
    }
    if (rc == DDN_IQ_OK) {
        const size_t a = align_of(caps[0]->info.sample_format);
        const size_t row = (size_t)n_min * a;
        char* buf = (char*)malloc(row * (size_t)n_captures);
        if (!buf) {
            rc = DDN_IQ_ERR_ALLOC;
        }
        for (int i = 0; i < n_captures && rc == DDN_IQ_OK; i++) {
            size_t got = 0;
            rc = ddn_iq_capture_read(caps[i], buf + (size_t)i * row, row, &got);
            if (rc == DDN_IQ_OK && got != row) {
                rc = DDN_IQ_ERR_IO;
            }
            if (rc == DDN_IQ_OK && caps[i]->info.sample_format == DDN_IQ_FORMAT_CU8 && caps[i]->info.fs4_shift_enabled) {
                fs4_rotate_cu8((unsigned char*)buf + (size_t)i * row, (size_t)n_min);
            }
        }
        if (rc == DDN_IQ_OK) {
            *out_buf = buf;
            *out_n_samples = (size_t)n_min;
            if (out_info0) {
                *out_info0 = caps[0]->info;
            }
        } else {
            free(buf);
        }
    }
    for (int i = 0; i < n_captures; i++) {
        ddn_iq_capture_close(caps[i]);
    }
    free(caps);
    return rc;
}

void
ddn_iq_free(void* p) {
    free(p);
}
