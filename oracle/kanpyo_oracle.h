/*
 * oracle/kanpyo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference hot path
 *   kanpyo::Tokenizer::tokenize()  (reference src/tokenizer.rs:16-45)
 * and of the dictionary pieces it reads.  It is the checker for the HIP path;
 * nothing in the product (kanpyo_amd/, include/) may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: the trie search, dup expansion and connection-matrix layout
 * are pinned by the reference's own known-answer tests (tests/golden/ JSON files,
 * copied as data from kanpyo-dict/src/trie/da.rs:252-351, index.rs:91-150,
 * connection.rs:58-71, builder/matrix_def.rs:70-85).  The reference asserts NO
 * Viterbi / token values anywhere (src/tests.rs:110-202 checks existence
 * only) and it cannot be compiled here (no cargo/rustc, un-vendored crates),
 * so token-level parity is "parity unpinned": the expected tokens in
 * tests/golden/fixture_tokens.json are hand-derived from the reference code
 * (SURVEY.md App. C), not produced by the reference binary.
 */
#ifndef KANPYO_ORACLE_H
#define KANPYO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct korc_dict korc_dict;

/* Token record; same 24-byte layout as include/kanpyo_gpu.h kgpu_token so the
 * tests can memcmp the two streams.  (reference src/token.rs:10-18; surface =
 * input[position .. position+byte_len], or the literal "EOS" for class 0.) */
typedef struct {
    int32_t id;        /* Token.id   (0 for the EOS dummy)            */
    uint32_t cls;      /* TokenClass: 0 Dummy, 1 Known, 2 Unknown      */
    uint32_t position; /* Token.position (byte offset in the sentence) */
    uint32_t start;    /* Token.start (char index)                     */
    uint32_t end;      /* Token.end   (char index; EOS: start+3)       */
    uint32_t byte_len; /* surface byte length (0 for EOS)              */
} korc_token;

/* Work counters of SURVEY.md section 8(d). */
typedef struct {
    uint64_t sentences;
    uint64_t B; /* input bytes                                         */
    uint64_t C; /* input chars                                         */
    uint64_t T; /* double-array byte steps attempted (incl. failing)   */
    uint64_t N; /* lattice nodes excluding BOS                         */
    uint64_t E; /* Viterbi relaxations                                 */
    uint64_t K; /* emitted tokens (incl. EOS)                          */
} korc_counters;

#define KORC_PANIC (-2)    /* the reference would panic (index out of bounds) */
#define KORC_CAPACITY (-3) /* caller's token buffer too small                  */

const char *korc_last_error(void);

/* Dictionary from the serialised blobs of SURVEY.md App. B (what
 * DictReadWrite::write_dict emits): index.dict, connection.dict, morph.dict,
 * the fixed-width prefix of unk.dict (trailing feature table ignored), plus the
 * three CharCategoryDef vectors raw. */
korc_dict *korc_dict_from_blobs(const uint8_t *index_dict, size_t index_len,
                                const uint8_t *connection_dict, size_t connection_len,
                                const uint8_t *morph_dict, size_t morph_len,
                                const uint8_t *unk_dict, size_t unk_len,
                                const uint8_t *char_category, size_t char_category_len,
                                const uint8_t *invoke_list, size_t invoke_len,
                                const uint8_t *group_list, size_t group_len);
void korc_dict_free(korc_dict *d);

/* Tokenizer::tokenize (src/tokenizer.rs:16-45).  Returns the token count
 * (0 is legal: unreachable EOS), KORC_PANIC or KORC_CAPACITY. */
int64_t korc_tokenize(const korc_dict *d, const uint8_t *utf8, size_t len,
                      korc_token *out, size_t cap, korc_counters *ctr);

/* Batch form used by the tests / CPU baseline: sentence i is
 * utf8[offsets[i]..offsets[i+1]).  Tokens are written densely; tok_offsets has
 * n+1 entries.  nthreads > 1 splits the sentence range contiguously over
 * pthreads; every sentence is tokenized exactly once (one thread writes straight
 * into `out`, several write into per-thread worst-case buffers that are copied
 * into place once the totals are known).  Returns 0 or a negative error. */
int korc_tokenize_batch(const korc_dict *d, const uint8_t *utf8, const uint64_t *offsets,
                        uint64_t n, korc_token *out, uint64_t cap, uint64_t *tok_offsets,
                        int nthreads, korc_counters *ctr);

/* All-core timing form (bench.py's cpu_baseline.all_cores): sentence i writes its tokens to
 * out[offsets[i] - offsets[0] + i ..] (B_i + 1 slots, always enough) and its count to tok_count[i]; threads
 * claim runs of 64 sentences from one counter and go over the corpus `reps` times.  No allocation, no
 * merge copy inside the call.  `out` needs (offsets[n] - offsets[0]) + n slots. */
int korc_tokenize_slots(const korc_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                        korc_token *out, uint32_t *tok_count, int nthreads, int reps, korc_counters *ctr);

/* IndexTable::search_common_prefix_of (kanpyo-dict/src/index.rs:40-53) for the
 * known-answer tests: writes up to cap (id, byte_len) pairs, returns the count
 * (0 == None). */
int64_t korc_common_prefix(const korc_dict *d, const uint8_t *utf8, size_t len,
                           int64_t *ids, uint64_t *lens, size_t cap);
/* DoubleArray::search (kanpyo-dict/src/trie/da.rs:133-153); 0 == None. */
int64_t korc_da_search(const uint8_t *index_dict, size_t index_len, const uint8_t *key, size_t len);

/* IndexTable::build (kanpyo-dict/src/index.rs:16-38) over the reference's
 * first-fit double-array builder (kanpyo-dict/src/trie/da.rs:22-131,191-217).
 * keys: n sorted (duplicates adjacent) keywords, concatenated, key i =
 * keys[key_offsets[i]..key_offsets[i+1]).  Returns a malloc'd index.dict blob
 * (free with korc_free). */
uint8_t *korc_index_build(const uint8_t *keys, const uint64_t *key_offsets, uint64_t n,
                          size_t *out_len);
void korc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
