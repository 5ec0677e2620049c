/*
 * oracle/kanpyo_oracle.c -- TEST INFRASTRUCTURE ONLY (see kanpyo_oracle.h).
 *
 * Plain-C CPU restatement of the reference algorithm.  Every function cites the
 * reference file:line it follows (paths relative to the reference checkout).
 * Token-level parity with the real reference binary is UNPINNED (header).
 */
#define _GNU_SOURCE
#include "kanpyo_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KORC_INVALID_UTF8 (-4)

static __thread char g_err[256];
const char *korc_last_error(void) { return g_err; }
static void set_err(const char *m) { snprintf(g_err, sizeof g_err, "%s", m); }

/* ------------------------------------------------------------------ types */

typedef struct { int32_t base, check; } da_node; /* trie/da.rs:13-17 */
typedef struct { int16_t left, right, cost; } morph_t; /* morph.rs:7-11 */

struct korc_dict {
    da_node *da; size_t da_len;                  /* trie/da.rs:19-20 DoubleArray(Vec<Node>) */
    int64_t *dup_key; uint64_t *dup_val; size_t dup_len; /* index.rs:12 BTreeMap */
    morph_t *morphs; size_t n_morphs;            /* morph.rs:24 */
    uint64_t row, col; int16_t *conn; size_t conn_len; /* connection.rs:5-9 */
    uint8_t *cat; size_t cat_len;                /* char_category_def.rs:17 */
    uint8_t *invoke; size_t invoke_len;          /* char_category_def.rs:18 */
    uint8_t *group; size_t group_len;            /* char_category_def.rs:19 */
    int unk_has[256]; int64_t unk_first[256]; uint64_t unk_count[256]; /* unk_dict.rs:15 */
    morph_t *unk_morphs; size_t n_unk_morphs;    /* unk_dict.rs:13 */
};

/* ------------------------------------------------------------ blob readers */

typedef struct { const uint8_t *p; size_t n, at; int bad; } rd_t;
static uint64_t rd_u64(rd_t *r) {
    if (r->at + 8 > r->n) { r->bad = 1; return 0; }
    uint64_t v; memcpy(&v, r->p + r->at, 8); r->at += 8; return v; /* LE host assumed */
}
static int32_t rd_i32(rd_t *r) {
    if (r->at + 4 > r->n) { r->bad = 1; return 0; }
    int32_t v; memcpy(&v, r->p + r->at, 4); r->at += 4; return v;
}
static int16_t rd_i16(rd_t *r) {
    if (r->at + 2 > r->n) { r->bad = 1; return 0; }
    int16_t v; memcpy(&v, r->p + r->at, 2); r->at += 2; return v;
}
static uint8_t rd_u8(rd_t *r) {
    if (r->at + 1 > r->n) { r->bad = 1; return 0; }
    return r->p[r->at++];
}

/* morph.rs:74-92 Morphs::from_dict: i64 n; n x (i16 left, i16 right, i16 cost) */
static morph_t *read_morphs(rd_t *r, size_t *n_out) {
    int64_t n = (int64_t)rd_u64(r);
    if (r->bad || n < 0 || (uint64_t)n > (r->n - r->at) / 6) { r->bad = 1; return NULL; }
    morph_t *m = (morph_t *)malloc(sizeof(morph_t) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        m[i].left = rd_i16(r); m[i].right = rd_i16(r); m[i].cost = rd_i16(r);
    }
    *n_out = (size_t)n;
    return m;
}

/* trie/da.rs:220-236 DoubleArray::from_dict + index.rs:57-73 IndexTable::from_dict */
static int read_index(rd_t *r, da_node **da, size_t *da_len, int64_t **dk, uint64_t **dv, size_t *dn) {
    uint64_t n = rd_u64(r);
    if (r->bad || n > (r->n - r->at) / 8) return -1;
    da_node *a = (da_node *)malloc(sizeof(da_node) * (size_t)(n ? n : 1));
    for (uint64_t i = 0; i < n; i++) { a[i].base = rd_i32(r); a[i].check = rd_i32(r); }
    uint64_t m = rd_u64(r);
    if (r->bad || m > (r->n - r->at) / 16) { free(a); return -1; }
    int64_t *k = (int64_t *)malloc(8 * (size_t)(m ? m : 1));
    uint64_t *v = (uint64_t *)malloc(8 * (size_t)(m ? m : 1));
    for (uint64_t i = 0; i < m; i++) { k[i] = (int64_t)rd_u64(r); v[i] = rd_u64(r); }
    if (r->bad) { free(a); free(k); free(v); return -1; }
    /* BTreeMap iteration order is ascending by key; write_dict emits that order
     * (index.rs:75-84).  Insert-sort defensively so lookup can binary-search. */
    for (uint64_t i = 1; i < m; i++) {
        int64_t kk = k[i]; uint64_t vv = v[i]; uint64_t j = i;
        while (j > 0 && k[j - 1] > kk) { k[j] = k[j - 1]; v[j] = v[j - 1]; j--; }
        k[j] = kk; v[j] = vv;
    }
    *da = a; *da_len = (size_t)n; *dk = k; *dv = v; *dn = (size_t)m;
    return 0;
}

korc_dict *korc_dict_from_blobs(const uint8_t *index_dict, size_t index_len,
                                const uint8_t *connection_dict, size_t connection_len,
                                const uint8_t *morph_dict, size_t morph_len,
                                const uint8_t *unk_dict, size_t unk_len,
                                const uint8_t *char_category, size_t char_category_len,
                                const uint8_t *invoke_list, size_t invoke_len,
                                const uint8_t *group_list, size_t group_len) {
    korc_dict *d = (korc_dict *)calloc(1, sizeof *d);
    rd_t r = {index_dict, index_len, 0, 0};
    if (read_index(&r, &d->da, &d->da_len, &d->dup_key, &d->dup_val, &d->dup_len)) {
        set_err("index.dict: truncated"); korc_dict_free(d); return NULL;
    }
    /* connection.rs:28-42: u64 row; u64 col; row*col x i16 */
    rd_t c = {connection_dict, connection_len, 0, 0};
    d->row = rd_u64(&c); d->col = rd_u64(&c);
    if (c.bad || (d->row && d->col > (c.n - c.at) / 2 / d->row)) {
        set_err("connection.dict: truncated"); korc_dict_free(d); return NULL;
    }
    d->conn_len = (size_t)(d->row * d->col);
    d->conn = (int16_t *)malloc(2 * (d->conn_len ? d->conn_len : 1));
    for (size_t i = 0; i < d->conn_len; i++) d->conn[i] = rd_i16(&c);
    rd_t m = {morph_dict, morph_len, 0, 0};
    d->morphs = read_morphs(&m, &d->n_morphs);
    if (m.bad) { set_err("morph.dict: truncated"); korc_dict_free(d); return NULL; }
    /* unk_dict.rs:75-99: u64 k; k x (u8 cat, i64 first_id, u64 count); Morphs; features.. */
    rd_t u = {unk_dict, unk_len, 0, 0};
    uint64_t k = rd_u64(&u);
    if (u.bad || k > (u.n - u.at) / 17) { set_err("unk.dict: truncated"); korc_dict_free(d); return NULL; }
    for (uint64_t i = 0; i < k; i++) {
        uint8_t cat = rd_u8(&u);
        int64_t first = (int64_t)rd_u64(&u);
        uint64_t cnt = rd_u64(&u);
        d->unk_has[cat] = 1; d->unk_first[cat] = first; d->unk_count[cat] = cnt;
    }
    d->unk_morphs = read_morphs(&u, &d->n_unk_morphs);
    if (u.bad) { set_err("unk.dict: truncated"); korc_dict_free(d); return NULL; }
    d->cat = (uint8_t *)malloc(char_category_len ? char_category_len : 1);
    memcpy(d->cat, char_category, char_category_len); d->cat_len = char_category_len;
    d->invoke = (uint8_t *)malloc(invoke_len ? invoke_len : 1);
    memcpy(d->invoke, invoke_list, invoke_len); d->invoke_len = invoke_len;
    d->group = (uint8_t *)malloc(group_len ? group_len : 1);
    memcpy(d->group, group_list, group_len); d->group_len = group_len;
    return d;
}

void korc_dict_free(korc_dict *d) {
    if (!d) return;
    free(d->da); free(d->dup_key); free(d->dup_val); free(d->morphs); free(d->conn);
    free(d->cat); free(d->invoke); free(d->group); free(d->unk_morphs); free(d);
}
void korc_free(void *p) { free(p); }

/* ------------------------------------------------- double-array trie search */

/* trie/da.rs:155-182 DoubleArray::search_common_prefix_of.  Signed i32 index
 * math; an index cast to usize that is negative or >= len is "None".
 * Returns count, or KORC_PANIC where the reference indexes out of bounds
 * (self.0[prev] with a 1-element array, da.rs:161). */
static int64_t da_common_prefix(const da_node *da, size_t len, const uint8_t *s, size_t n,
                                int64_t *ids, uint64_t *lens, size_t cap, uint64_t *steps) {
    int32_t p = 1; /* ROOT_ID da.rs:9,156 */
    int64_t cnt = 0;
    for (size_t i = 0; i < n; i++) {
        int32_t prev = p;
        if ((size_t)prev >= len) return KORC_PANIC;  /* self.0[prev as usize] da.rs:161 */
        p = da[prev].base + (int32_t)s[i];
        if (steps) (*steps)++;
        if (p < 0 || (size_t)p >= len || da[p].check != prev) break; /* da.rs:162-165 */
        int32_t ahead = da[p].base + 0;                              /* TERMINATOR da.rs:166 */
        if (ahead >= 0 && (size_t)ahead < len && da[ahead].check == p && da[ahead].base < 0) {
            if ((size_t)cnt < cap) { ids[cnt] = -(int64_t)da[ahead].base; lens[cnt] = i + 1; }
            cnt++;                                                   /* da.rs:168-175 */
        }
    }
    return cnt;
}

/* trie/da.rs:133-153 DoubleArray::search */
int64_t korc_da_search(const uint8_t *index_dict, size_t index_len, const uint8_t *key, size_t n) {
    rd_t r = {index_dict, index_len, 0, 0};
    da_node *da; size_t len; int64_t *dk; uint64_t *dv; size_t dn;
    if (read_index(&r, &da, &len, &dk, &dv, &dn)) return 0;
    int64_t out = 0;
    int32_t p = 1;
    for (size_t i = 0; i < n; i++) {
        if (p < 0 || (size_t)p >= len) goto done;   /* self.0.get(p)? */
        int32_t q = da[p].base + (int32_t)key[i];
        if (q < 0 || (size_t)q >= len) goto done;
        if (da[q].check != p) goto done;
        p = q;
    }
    if (p < 0 || (size_t)p >= len) goto done;
    {
        int32_t q = da[p].base + 0 + 0;
        if (q < 0 || (size_t)q >= len) goto done;
        if (da[q].check == p) out = -(int64_t)da[q].base;
    }
done:
    free(da); free(dk); free(dv);
    return out;
}

static uint64_t dup_of(const korc_dict *d, int64_t id) { /* index.rs:47 dup.get(id).unwrap_or(0) */
    size_t lo = 0, hi = d->dup_len;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (d->dup_key[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < d->dup_len && d->dup_key[lo] == id) ? d->dup_val[lo] : 0;
}

/* index.rs:40-53 IndexTable::search_common_prefix_of */
int64_t korc_common_prefix(const korc_dict *d, const uint8_t *utf8, size_t len,
                           int64_t *ids, uint64_t *lens, size_t cap) {
    int64_t raw_ids[256]; uint64_t raw_lens[256];
    int64_t m = da_common_prefix(d->da, d->da_len, utf8, len, raw_ids, raw_lens, 256, NULL);
    if (m < 0) return m;
    if (m > 256) m = 256;
    int64_t cnt = 0;
    for (int64_t k = 0; k < m; k++) {
        uint64_t dup = dup_of(d, raw_ids[k]);
        for (uint64_t i = 0; i <= dup; i++) {
            if ((size_t)cnt < cap) { ids[cnt] = raw_ids[k] + (int64_t)i; lens[cnt] = raw_lens[k]; }
            cnt++;
        }
    }
    return cnt;
}

/* ---------------------------------------------------------------- lattice */

typedef struct {
    int32_t id;        /* lattice/node.rs:7,27-32: 0 for Dummy */
    uint32_t byte_pos; /* node.rs:8 */
    uint32_t char_pos; /* node.rs:9 */
    uint32_t byte_len; /* surface.len() */
    uint32_t char_len; /* surface.chars().count() */
    morph_t morph;     /* node.rs:10 */
    uint8_t cls;       /* 0 Dummy, 1 Known, 2 Unknown (node.rs:15-23) */
} lnode;

typedef struct {
    /* chars */
    uint32_t *cp; uint32_t *cbyte; size_t ccap;
    /* Lattice.nodes (lattice.rs:8) */
    lnode *nodes; size_t n_nodes, ncap;
    /* Lattice.edges (lattice.rs:9): per end position an insertion-ordered list,
     * kept as head/tail/next so that push is O(1) and iteration order equals the
     * Vec<usize> push order. */
    uint32_t *ehead, *etail; size_t ecap;
    uint32_t *enext;
    int32_t *dp; uint8_t *dp_some; int64_t *pre; /* lattice.rs:118-119 */
    uint32_t *path; size_t pcap;
} ws_t;

#define NONE32 0xFFFFFFFFu

static void ws_free(ws_t *w) {
    free(w->cp); free(w->cbyte); free(w->nodes); free(w->ehead); free(w->etail);
    free(w->enext); free(w->dp); free(w->dp_some); free(w->pre); free(w->path);
    memset(w, 0, sizeof *w);
}
static void ws_chars(ws_t *w, size_t c) {
    if (c + 2 <= w->ccap) return;
    w->ccap = (c + 2) * 2;
    w->cp = (uint32_t *)realloc(w->cp, 4 * w->ccap);
    w->cbyte = (uint32_t *)realloc(w->cbyte, 4 * w->ccap);
    w->ehead = (uint32_t *)realloc(w->ehead, 4 * w->ccap);
    w->etail = (uint32_t *)realloc(w->etail, 4 * w->ccap);
    w->path = (uint32_t *)realloc(w->path, 4 * w->ccap);
}
static lnode *ws_push(ws_t *w) {
    if (w->n_nodes == w->ncap) {
        w->ncap = w->ncap ? w->ncap * 2 : 1024;
        w->nodes = (lnode *)realloc(w->nodes, sizeof(lnode) * w->ncap);
        w->enext = (uint32_t *)realloc(w->enext, 4 * w->ncap);
        w->dp = (int32_t *)realloc(w->dp, 4 * w->ncap);
        w->dp_some = (uint8_t *)realloc(w->dp_some, w->ncap);
        w->pre = (int64_t *)realloc(w->pre, 8 * w->ncap);
    }
    return &w->nodes[w->n_nodes++];
}
static void edge_push(ws_t *w, size_t end, uint32_t idx) { /* self.edges[end].push(idx) */
    w->enext[idx] = NONE32;
    if (w->ehead[end] == NONE32) w->ehead[end] = idx; else w->enext[w->etail[end]] = idx;
    w->etail[end] = idx;
}

/* Rust's str is always valid UTF-8; the C boundary has to check. */
static int64_t decode_utf8(const uint8_t *s, size_t n, ws_t *w) {
    ws_chars(w, n);
    size_t c = 0, i = 0;
    while (i < n) {
        uint32_t b0 = s[i], cp; size_t l;
        if (b0 < 0x80) { cp = b0; l = 1; }
        else if (b0 >= 0xC2 && b0 <= 0xDF) { cp = b0 & 0x1F; l = 2; }
        else if (b0 >= 0xE0 && b0 <= 0xEF) { cp = b0 & 0x0F; l = 3; }
        else if (b0 >= 0xF0 && b0 <= 0xF4) { cp = b0 & 0x07; l = 4; }
        else return KORC_INVALID_UTF8;
        if (i + l > n) return KORC_INVALID_UTF8;
        for (size_t k = 1; k < l; k++) {
            uint32_t b = s[i + k];
            if ((b & 0xC0) != 0x80) return KORC_INVALID_UTF8;
            cp = (cp << 6) | (b & 0x3F);
        }
        if (l == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return KORC_INVALID_UTF8;
        if (l == 4 && (cp < 0x10000 || cp > 0x10FFFF)) return KORC_INVALID_UTF8;
        w->cp[c] = cp; w->cbyte[c] = (uint32_t)i; c++;
        i += l;
    }
    w->cbyte[c] = (uint32_t)n;
    return (int64_t)c;
}

/* char_category_def.rs:33-38 CharCategoryDef::char_category */
static int char_category(const korc_dict *d, uint32_t ch, uint8_t *out) {
    if ((size_t)ch < d->cat_len) { *out = d->cat[ch]; return 0; }
    if (d->cat_len == 0) return -1; /* &self.char_category[0] panics */
    *out = d->cat[0];
    return 0;
}

/* lattice.rs:101-114 Lattice::build + lattice.rs:116-154 Lattice::viterbi +
 * tokenizer.rs:16-45 Tokenizer::tokenize */
static int64_t tokenize_ws(const korc_dict *d, const uint8_t *s, size_t n, korc_token *out,
                           size_t cap, korc_counters *ctr, ws_t *w) {
    int64_t C64 = decode_utf8(s, n, w);
    if (C64 < 0) { set_err("invalid UTF-8"); return C64; }
    size_t C = (size_t)C64;
    uint64_t T = 0, E = 0;

    /* Lattice::new lattice.rs:13-20: edges = vec![vec![]; chars + 2] */
    for (size_t e = 0; e < C + 2; e++) w->ehead[e] = w->etail[e] = NONE32;
    w->n_nodes = 0;

    /* add_bos_node lattice.rs:156-164 */
    {
        lnode *b = ws_push(w);
        memset(b, 0, sizeof *b);
        edge_push(w, 0, 0);
    }

    for (size_t char_pos = 0; char_pos < C; char_pos++) { /* lattice.rs:105 */
        size_t byte_pos = w->cbyte[char_pos];
        /* process_known_words lattice.rs:24-38 */
        int matched_known = 0;
        {
            int64_t ids[256]; uint64_t lens[256];
            int64_t m = da_common_prefix(d->da, d->da_len, s + byte_pos, n - byte_pos, ids, lens, 256, &T);
            if (m == KORC_PANIC) { set_err("panic: double array index out of bounds"); return KORC_PANIC; }
            if (m > 256) { set_err("oracle: more than 256 prefix matches"); return KORC_PANIC; }
            matched_known = m > 0; /* Some(..) iff non-empty, da.rs:177-181 */
            for (int64_t k = 0; k < m; k++) {
                uint64_t dup = dup_of(d, ids[k]);
                for (uint64_t i = 0; i <= dup; i++) { /* index.rs:46-51 */
                    int64_t id = ids[k] + (int64_t)i;
                    size_t end_byte = byte_pos + lens[k];
                    /* &input[byte_pos..end_byte_pos] lattice.rs:32: must be a char boundary */
                    if (end_byte < n && (s[end_byte] & 0xC0) == 0x80) {
                        set_err("panic: match ends inside a char"); return KORC_PANIC;
                    }
                    /* add_known_node lattice.rs:177-188 */
                    if (id - 1 < 0 || (size_t)(id - 1) >= d->n_morphs) {
                        set_err("panic: morphs[id-1] out of bounds"); return KORC_PANIC;
                    }
                    size_t clen = 0;
                    for (size_t b = byte_pos; b < end_byte; b++) clen += (s[b] & 0xC0) != 0x80;
                    lnode *nd = ws_push(w);
                    nd->id = (int32_t)id; nd->byte_pos = (uint32_t)byte_pos;
                    nd->char_pos = (uint32_t)char_pos; nd->byte_len = (uint32_t)lens[k];
                    nd->char_len = (uint32_t)clen; nd->morph = d->morphs[id - 1]; nd->cls = 1;
                    edge_push(w, char_pos + clen, (uint32_t)(w->n_nodes - 1));
                }
            }
        }
        /* process_unknown_words lattice.rs:42-99 */
        {
            uint8_t cat;
            if (char_category(d, w->cp[char_pos], &cat)) { set_err("panic: empty char table"); return KORC_PANIC; }
            int gate = !matched_known;
            if (!gate) { /* short-circuit ||: invoke_list is indexed only when matched_known */
                if ((size_t)cat >= d->invoke_len) { set_err("panic: invoke_list index"); return KORC_PANIC; }
                gate = d->invoke[cat] != 0;
            }
            if (gate) { /* lattice.rs:54 */
                int is_group = (size_t)cat < d->group_len ? d->group[cat] != 0 : 0; /* :58-63 */
                size_t end_char = char_pos + 1, ulen = 1;
                if (is_group) { /* lattice.rs:70-84 */
                    while (end_char < C) {
                        uint8_t nc;
                        char_category(d, w->cp[end_char], &nc);
                        if (nc != cat) break;
                        end_char++; ulen++;
                        if (ulen >= 1024) break; /* MAXIMUM_UNKNOWN_WORD_LENGTH lattice.rs:55,80 */
                    }
                }
                if (d->unk_has[cat]) { /* lattice.rs:87-97 */
                    for (uint64_t i = 0; i < d->unk_count[cat]; i++) {
                        int64_t id = d->unk_first[cat] + (int64_t)i;
                        /* add_unknown_node lattice.rs:190-201 */
                        if (id - 1 < 0 || (size_t)(id - 1) >= d->n_unk_morphs) {
                            set_err("panic: unk morphs[id-1] out of bounds"); return KORC_PANIC;
                        }
                        lnode *nd = ws_push(w);
                        nd->id = (int32_t)id; nd->byte_pos = (uint32_t)byte_pos;
                        nd->char_pos = (uint32_t)char_pos;
                        nd->byte_len = w->cbyte[end_char] - (uint32_t)byte_pos;
                        nd->char_len = (uint32_t)ulen; nd->morph = d->unk_morphs[id - 1]; nd->cls = 2;
                        edge_push(w, char_pos + ulen, (uint32_t)(w->n_nodes - 1));
                    }
                }
            }
        }
    }
    /* add_eos_node lattice.rs:165-175 */
    {
        lnode *e = ws_push(w);
        memset(e, 0, sizeof *e);
        e->byte_pos = (uint32_t)n; e->char_pos = (uint32_t)C;
        edge_push(w, C + 1, (uint32_t)(w->n_nodes - 1));
    }

    /* viterbi lattice.rs:116-142 */
    const int32_t INF = 1 << 30;
    size_t N = w->n_nodes;
    for (size_t i = 0; i < N; i++) { w->dp_some[i] = 0; w->dp[i] = 0; w->pre[i] = -1; }
    for (size_t char_pos = 1; char_pos < C + 2; char_pos++) {
        for (uint32_t i = w->ehead[char_pos]; i != NONE32; i = w->enext[i]) {
            const lnode *target = &w->nodes[i];
            w->dp[i] = INF; w->dp_some[i] = 1;
            size_t tpos = target->char_pos;
            for (uint32_t j = w->ehead[tpos]; j != NONE32; j = w->enext[j]) {
                const lnode *previous = &w->nodes[j];
                int32_t prev_cost = w->dp_some[j] ? w->dp[j] : 0; /* unwrap_or(0) */
                int32_t cost = (int32_t)target->morph.cost;
                /* ConnectionTable::get(row = prev.right_id, col = target.left_id) =
                 * data[self.row * col + row]  connection.rs:12-14; i16 as usize casts */
                uint64_t r = (uint64_t)(int64_t)previous->morph.right;
                uint64_t c = (uint64_t)(int64_t)target->morph.left;
                uint64_t idx = d->row * c + r;
                if (idx >= d->conn_len) { set_err("panic: connection index out of bounds"); return KORC_PANIC; }
                int32_t matrix_cost = (int32_t)d->conn[idx];
                int32_t total = prev_cost + cost + matrix_cost;
                if (total > INF) total = INF; /* .min(INF) */
                E++;
                if (total < w->dp[i]) { w->dp[i] = total; w->pre[i] = (int64_t)j; } /* strict */
            }
        }
    }
    /* backtrace lattice.rs:144-153 */
    size_t pos = N - 1, K = 0;
    while (w->pre[pos] >= 0) {
        w->path[K++] = (uint32_t)pos;
        pos = (size_t)w->pre[pos];
    }
    if (ctr) {
        ctr->sentences += 1; ctr->B += n; ctr->C += C; ctr->T += T;
        ctr->N += N - 1; ctr->E += E; ctr->K += K;
    }
    if (K > cap) return KORC_CAPACITY;
    /* tokenizer.rs:22-43 map Node -> Token, in forward order (paths.reverse()) */
    for (size_t k = 0; k < K; k++) {
        const lnode *nd = &w->nodes[w->path[K - 1 - k]];
        korc_token *t = &out[k];
        t->id = nd->id; t->cls = nd->cls; t->position = nd->byte_pos; t->start = nd->char_pos;
        if (nd->cls == 0) { t->end = nd->char_pos + 3; t->byte_len = 0; } /* "EOS".chars().count() */
        else { t->end = nd->char_pos + nd->char_len; t->byte_len = nd->byte_len; }
    }
    return (int64_t)K;
}

int64_t korc_tokenize(const korc_dict *d, const uint8_t *utf8, size_t len, korc_token *out,
                      size_t cap, korc_counters *ctr) {
    ws_t w; memset(&w, 0, sizeof w);
    int64_t r = tokenize_ws(d, utf8, len, out, cap, ctr, &w);
    ws_free(&w);
    return r;
}

/* ------------------------------------------------------------------ batch */

/* One pass: every sentence is tokenized exactly once.  A thread owns a contiguous sentence range and
 * writes its tokens densely -- straight into the caller's buffer when it is the only thread, else into
 * a scratch buffer of worst-case size (tokens <= chars + 1 <= bytes + 1 per sentence) that is copied to
 * its final place once the per-thread totals are known. */
typedef struct {
    const korc_dict *d; const uint8_t *utf8; const uint64_t *off; uint64_t lo, hi;
    korc_token *buf; uint64_t cap, used; /* dense tokens of [lo, hi) */
    uint64_t *tok_off;                   /* tok_off[i + 1] = tokens of sentence i (prefix-summed later) */
    korc_counters ctr; int err;
} job_t;

static void *job_run(void *arg) {
    job_t *j = (job_t *)arg;
    ws_t w; memset(&w, 0, sizeof w);
    for (uint64_t i = j->lo; i < j->hi; i++) {
        size_t n = (size_t)(j->off[i + 1] - j->off[i]);
        int64_t k = tokenize_ws(j->d, j->utf8 + j->off[i], n, j->buf + j->used, (size_t)(j->cap - j->used), &j->ctr, &w);
        if (k < 0) { if (k != KORC_INVALID_UTF8 || !j->err) j->err = (int)k; if (k == KORC_CAPACITY) break; k = 0; }
        j->tok_off[i + 1] = (uint64_t)k;
        j->used += (uint64_t)k;
    }
    ws_free(&w);
    return NULL;
}

int korc_tokenize_batch(const korc_dict *d, const uint8_t *utf8, const uint64_t *offsets,
                        uint64_t n, korc_token *out, uint64_t cap, uint64_t *tok_offsets,
                        int nthreads, korc_counters *ctr) {
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > n) nthreads = n ? (int)n : 1;
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    int err = 0;
    tok_offsets[0] = 0;
    for (int t = 0; t < nthreads; t++) {
        job_t *j = &jobs[t];
        j->d = d; j->utf8 = utf8; j->off = offsets; j->tok_off = tok_offsets;
        j->lo = n * (uint64_t)t / (uint64_t)nthreads; j->hi = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
        if (nthreads == 1) { j->buf = out; j->cap = cap; }
        else {
            j->cap = (offsets[j->hi] - offsets[j->lo]) + (j->hi - j->lo);
            j->buf = (korc_token *)malloc(sizeof(korc_token) * (size_t)(j->cap ? j->cap : 1));
        }
        if (nthreads == 1) job_run(j); else pthread_create(&th[t], NULL, job_run, j);
    }
    uint64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        if (jobs[t].err && jobs[t].err != KORC_INVALID_UTF8) err = jobs[t].err;
        total += jobs[t].used;
    }
    if (!err && total > cap) err = KORC_CAPACITY;
    if (!err) {
        for (uint64_t i = 0; i < n; i++) tok_offsets[i + 1] += tok_offsets[i];
        if (nthreads > 1)
            for (int t = 0; t < nthreads; t++)
                memcpy(out + tok_offsets[jobs[t].lo], jobs[t].buf, sizeof(korc_token) * (size_t)jobs[t].used);
    }
    if (ctr) for (int t = 0; t < nthreads; t++) {
        ctr->sentences += jobs[t].ctr.sentences; ctr->B += jobs[t].ctr.B; ctr->C += jobs[t].ctr.C;
        ctr->T += jobs[t].ctr.T; ctr->N += jobs[t].ctr.N; ctr->E += jobs[t].ctr.E; ctr->K += jobs[t].ctr.K;
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) free(jobs[t].buf);
    free(jobs); free(th);
    return err;
}

/* All-core timing form: nothing but tokenization inside the threads.  Sentence i owns the token slots
 * out[offsets[i] - offsets[0] + i ..] (tokens <= chars + 1 <= bytes + 1, so B_i + 1 slots always suffice: disjoint
 * ranges known up front, no per-thread buffers, no merge copy, no allocator); tok_count[i] = its token count.
 * Threads claim runs of 64 sentences from one counter and go over the corpus `reps` times (thread start-up is paid
 * once per call).  The reference hands every caller its own Vec<Token> (src/tokenizer.rs:16-45): per-sentence
 * slots are the closest flat equivalent. */
typedef struct {
    const korc_dict *d; const uint8_t *utf8; const uint64_t *off; uint64_t n;
    korc_token *out; uint32_t *tok_count; int reps;
    volatile uint64_t *next; korc_counters ctr; int err;
} slot_job_t;

static void *slot_job_run(void *arg) {
    slot_job_t *j = (slot_job_t *)arg;
    ws_t w; memset(&w, 0, sizeof w);
    const uint64_t total = j->n * (uint64_t)j->reps, RUN = 64;
    for (;;) {
        uint64_t g = __atomic_fetch_add(j->next, RUN, __ATOMIC_RELAXED);
        if (g >= total) break;
        uint64_t ge = g + RUN < total ? g + RUN : total;
        for (; g < ge; g++) {
            uint64_t i = g % j->n;
            size_t nb = (size_t)(j->off[i + 1] - j->off[i]);
            korc_token *dst = j->out + (j->off[i] - j->off[0]) + i;
            int64_t k = tokenize_ws(j->d, j->utf8 + j->off[i], nb, dst, nb + 1, &j->ctr, &w);
            if (k < 0) { if (k != KORC_INVALID_UTF8 || !j->err) j->err = (int)k; k = 0; }
            j->tok_count[i] = (uint32_t)k;
        }
    }
    ws_free(&w);
    return NULL;
}

int korc_tokenize_slots(const korc_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                        korc_token *out, uint32_t *tok_count, int nthreads, int reps, korc_counters *ctr) {
    if (nthreads < 1) nthreads = 1;
    if (reps < 1) reps = 1;
    slot_job_t *jobs = (slot_job_t *)calloc((size_t)nthreads, sizeof(slot_job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    volatile uint64_t next = 0;
    int err = 0;
    for (int t = 0; t < nthreads; t++) {
        slot_job_t *j = &jobs[t];
        j->d = d; j->utf8 = utf8; j->off = offsets; j->n = n; j->out = out; j->tok_count = tok_count; j->reps = reps; j->next = &next;
        if (n == 0) continue;
        if (nthreads == 1) slot_job_run(j); else pthread_create(&th[t], NULL, slot_job_run, j);
    }
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1 && n) pthread_join(th[t], NULL);
        if (jobs[t].err && jobs[t].err != KORC_INVALID_UTF8) err = jobs[t].err;
        if (ctr) {
            ctr->sentences += jobs[t].ctr.sentences; ctr->B += jobs[t].ctr.B; ctr->C += jobs[t].ctr.C;
            ctr->T += jobs[t].ctr.T; ctr->N += jobs[t].ctr.N; ctr->E += jobs[t].ctr.E; ctr->K += jobs[t].ctr.K;
        }
    }
    free(jobs); free(th);
    return err;
}

/* ----------------------------------------------------- double-array builder */

#define INIT_BUFFER_SIZE (50 * 1024) /* trie/da.rs:6 */
#define EXPAND_RATIO 2               /* trie/da.rs:7 */

typedef struct {
    da_node *a; size_t len;
    const uint8_t *keys; const uint64_t *koff; /* unique sorted keywords */
    const int64_t *ids;
} dab_t;

static void dab_expand(dab_t *b) { /* da.rs:37-41 */
    size_t nl = b->len * EXPAND_RATIO;
    b->a = (da_node *)realloc(b->a, sizeof(da_node) * nl);
    memset(b->a + b->len, 0, sizeof(da_node) * (nl - b->len));
    b->len = nl;
}

static size_t dab_seek(dab_t *b, const uint8_t *chars, size_t nch) { /* da.rs:43-78 */
    size_t left = (size_t)b->a[0].base;
    for (size_t i = left;; i++) {
        int found = 1;
        while (i >= b->len) dab_expand(b);
        for (size_t k = 0; k < nch; k++) {
            int32_t q = (int32_t)i + (int32_t)chars[k];
            while (q >= (int32_t)b->len) dab_expand(b);
            if (b->a[q].check != 0) { found = 0; break; }
        }
        if (found) {
            size_t used = 0;
            for (size_t x = left; x <= i; x++) if (b->a[x].check != 0) used++;
            double occupancy = (double)used / (double)(i - left + 1);
            if (occupancy >= 0.95) b->a[0].base = (int32_t)i + 1;
            return i;
        }
    }
}

static size_t key_len(const dab_t *b, uint64_t k) { return (size_t)(b->koff[k + 1] - b->koff[k]); }

/* da.rs:80-131 DoubleArray::add.  `branches` of the reference is always a
 * contiguous run [lo,hi) of the sorted unique keyword list (all keys sharing
 * the first i bytes), so the Vec<KeywordID> is carried as a range. */
static int dab_add(dab_t *b, size_t p, size_t i, uint64_t lo, uint64_t hi) {
    while (p >= b->len) dab_expand(b);
    uint8_t chars[257]; uint64_t cstart[257], cend[257]; size_t nch = 0;
    uint8_t seen[256]; memset(seen, 0, sizeof seen);
    for (uint64_t k = lo; k < hi; k++) {
        const uint8_t *s = b->keys + b->koff[k];
        uint8_t ch = i < key_len(b, k) ? s[i] : 0; /* *str.get(i).unwrap_or(&TERMINATOR) */
        if (nch == 0 || chars[nch - 1] != ch) {
            /* a byte re-appearing in a second run makes char_bytes hold it twice and the
             * assert at da.rs:106-111 fire; sorted keyword lists never do this */
            if (seen[ch]) return -1;
            seen[ch] = 1;
            chars[nch] = ch; cstart[nch] = k; cend[nch] = k; nch++;
        }
        cend[nch - 1] = k + 1;
    }
    size_t left = dab_seek(b, chars, nch);
    b->a[p].base = (int32_t)left;
    for (size_t c = 0; c < nch; c++) {
        int32_t q = (int32_t)left + (int32_t)chars[c];
        b->a[q].check = (int32_t)p;
        if (chars[c] == 0) b->a[q].base = -(int32_t)b->ids[lo]; /* leaf: ids[branches[0]] */
    }
    for (size_t c = 0; c < nch; c++) {
        if (chars[c] == 0) continue; /* TERMINATOR has no child branches */
        int32_t q = b->a[p].base + (int32_t)chars[c];
        if (dab_add(b, (size_t)q, i + 1, cstart[c], cend[c])) return -1;
    }
    return 0;
}

/* index.rs:16-38 IndexTable::build + da.rs:205-217 build_with_ids + da.rs:29-35
 * truncate + index.rs:75-84 / da.rs:237-245 write_dict */
uint8_t *korc_index_build(const uint8_t *keys, const uint64_t *key_offsets, uint64_t n,
                          size_t *out_len) {
    uint64_t *ukoff = (uint64_t *)malloc(8 * (size_t)(n + 1));
    uint8_t *ukeys = (uint8_t *)malloc((size_t)(key_offsets[n] - key_offsets[0]) + 1);
    int64_t *ids = (int64_t *)malloc(8 * (size_t)(n + 1));
    int64_t *dk = (int64_t *)malloc(8 * (size_t)(n + 1));
    uint64_t *dv = (uint64_t *)malloc(8 * (size_t)(n + 1));
    uint64_t nu = 0, nd = 0, at = 0;
    int have_prev = 0; uint64_t prev_i = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *s = keys + key_offsets[i]; size_t l = (size_t)(key_offsets[i + 1] - key_offsets[i]);
        if (have_prev) {
            const uint8_t *ps = keys + key_offsets[prev_i];
            size_t pl = (size_t)(key_offsets[prev_i + 1] - key_offsets[prev_i]);
            if (pl == l && memcmp(ps, s, l) == 0) {
                int64_t prev_no = (int64_t)prev_i + 1;
                if (nd && dk[nd - 1] == prev_no) dv[nd - 1]++; else { dk[nd] = prev_no; dv[nd] = 1; nd++; }
                continue;
            }
        }
        have_prev = 1; prev_i = i;
        ukoff[nu] = at; memcpy(ukeys + at, s, l); at += l;
        ids[nu] = (int64_t)i + 1; nu++;
    }
    ukoff[nu] = at;

    dab_t b; memset(&b, 0, sizeof b);
    b.len = INIT_BUFFER_SIZE;
    b.a = (da_node *)calloc(b.len, sizeof(da_node));
    b.a[0].base = 1 + 1; /* ROOT_ID + 1, da.rs:25 */
    b.keys = ukeys; b.koff = ukoff; b.ids = ids;
    if (dab_add(&b, 1, 0, 0, nu)) {
        set_err("panic: keywords not grouped by prefix (assert at da.rs:106)");
        free(b.a); free(ukoff); free(ukeys); free(ids); free(dk); free(dv);
        return NULL;
    }
    size_t len = b.len; /* truncate da.rs:29-35 */
    while (len > 1 && b.a[len - 1].check == 0) len--;

    size_t bytes = 8 + len * 8 + 8 + (size_t)nd * 16;
    uint8_t *blob = (uint8_t *)malloc(bytes);
    size_t o = 0;
    uint64_t u = len; memcpy(blob + o, &u, 8); o += 8;
    for (size_t i = 0; i < len; i++) { memcpy(blob + o, &b.a[i].base, 4); memcpy(blob + o + 4, &b.a[i].check, 4); o += 8; }
    u = nd; memcpy(blob + o, &u, 8); o += 8;
    for (uint64_t i = 0; i < nd; i++) { memcpy(blob + o, &dk[i], 8); memcpy(blob + o + 8, &dv[i], 8); o += 16; }
    *out_len = bytes;
    free(b.a); free(ukoff); free(ukeys); free(ids); free(dk); free(dv);
    return blob;
}
