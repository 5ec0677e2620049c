/*
 * include/kanpyo_gpu.h -- C ABI of libkanpyo_gpu.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of togatoga/kanpyo: Tokenizer::tokenize()
 * (lattice build over the double-array trie + Viterbi over the connection
 * matrix).  The reference has no FFI of its own (pure safe Rust); these are the
 * entry points a Rust `kanpyo::Tokenizer` shim binds with `extern "C"` (the
 * binding is shown in INTEGRATION.md).  Each entry point cites the reference
 * interface it replaces; paths are relative to the reference checkout.
 *
 * Plain pointers and sizes only; no torch / HIP types in the signatures (a HIP
 * stream crosses as void*).  All functions return KGPU_OK (0) or a KGPU_ERR_*
 * code and never abort; kgpu_last_error() gives the thread-local message.
 */
#ifndef KANPYO_GPU_H
#define KANPYO_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGPU_OK 0
#define KGPU_ERR_INVALID_ARG 1 /* null pointer, offsets not monotone, ...                */
#define KGPU_ERR_BAD_DICT 2    /* truncated blob, or a dictionary on which the reference
                                  itself would panic (index out of bounds)               */
#define KGPU_ERR_HIP 3         /* HIP runtime error (message has the hipError string)    */
#define KGPU_ERR_CAPACITY 4    /* caller's token buffer too small; *n_tokens = needed    */
#define KGPU_ERR_NO_DEVICE 5   /* no usable gfx950 device / extension built without one  */
#define KGPU_ERR_INTERNAL 6

/* Per-sentence status (uint8). */
#define KGPU_SENT_OK 0
#define KGPU_SENT_INVALID_UTF8 1 /* Rust's &str cannot carry this; the C boundary checks
                                    (reference src/tokenizer.rs:16 takes &str).  The
                                    sentence yields zero tokens.                        */
#define KGPU_SENT_NO_SCRATCH 2   /* transient: the lattice did not fit the scratch arena;
                                    kgpu_ctx_sync grows it and reruns the batch, callers
                                    never see this value                                 */
#define KGPU_SENT_TRUNCATED 3    /* measurement mode only (kgpu_ctx_set_ablation): the
                                    sentence was stopped after a stage, zero tokens      */

/* TokenClass (reference src/token.rs:3-8). */
#define KGPU_CLASS_DUMMY 0
#define KGPU_CLASS_KNOWN 1
#define KGPU_CLASS_UNKNOWN 2

/* Token (reference src/token.rs:10-18) as a fixed 24-byte record.  `surface`
 * is not materialised: it is input[position .. position + byte_len] of the
 * sentence, or the literal "EOS" when cls == KGPU_CLASS_DUMMY (byte_len == 0,
 * end == start + 3 because "EOS".chars().count() == 3, src/tokenizer.rs:28,34). */
typedef struct kgpu_token {
    int32_t id;        /* Token.id (KeywordID; 0 for the EOS dummy)      */
    uint32_t cls;      /* Token.class                                     */
    uint32_t position; /* Token.position: byte offset inside the sentence */
    uint32_t start;    /* Token.start: char index                         */
    uint32_t end;      /* Token.end: char index                           */
    uint32_t byte_len; /* surface length in bytes                         */
} kgpu_token;

/* The same Token in 8 bytes, for moving results over PCIe / xGMI (a third of the volume): consecutive tokens of a
 * sentence tile it (a path through the lattice: each word starts where the previous one ends, src/lattice.rs:144-153),
 * so position and start are running sums over the sentence and need not travel:
 *   packed = cls | chars << 2 | byte_len << 14     (chars = end - start, 12 bits; byte_len 18 bits; EOS: chars 3, byte_len 0)
 * plus, per sentence, the first token's (position, start) -- not always (0, 0): when the best chain starts at an
 * unreachable node that node is dropped (src/lattice.rs:144-153).  kgpu_expand_tokens restores
 * the 24-byte records exactly. */
typedef struct kgpu_token8 {
    int32_t id;
    uint32_t packed;
} kgpu_token8;
#define KGPU_T8_CLS(p) ((p) & 3u)
#define KGPU_T8_CHARS(p) (((p) >> 2) & 0xFFFu)
#define KGPU_T8_BYTES(p) ((p) >> 14)

/* The dictionary tables exactly as the reference serialises them
 * (DictReadWrite::write_dict, kanpyo-dict/src/dict.rs:13-18): the hot-path
 * table internals are private in the reference (trie/da.rs:14-20,
 * index.rs:10-13, connection.rs:5-9, morph.rs:24), write_dict is their only
 * public egress, so the shim hands over those bytes.
 *   index_dict      index.rs:75-84 + trie/da.rs:237-245
 *   connection_dict connection.rs:44-51
 *   morph_dict      morph.rs:61-72
 *   unk_dict        unk_dict.rs:60-73 (the trailing feature table is ignored)
 *   char_category / invoke_list / group_list: the pub Vec<u8>/Vec<bool> fields
 *                   of CharCategoryDef (char_category_def.rs:14-20), one byte each */
typedef struct kgpu_dict_blobs {
    const uint8_t *index_dict;      size_t index_len;
    const uint8_t *connection_dict; size_t connection_len;
    const uint8_t *morph_dict;      size_t morph_len;
    const uint8_t *unk_dict;        size_t unk_len;
    const uint8_t *char_category;   size_t char_category_len;
    const uint8_t *invoke_list;     size_t invoke_len;
    const uint8_t *group_list;      size_t group_len;
} kgpu_dict_blobs;

typedef struct kgpu_dict kgpu_dict; /* owns the HBM-resident tables           */
typedef struct kgpu_ctx kgpu_ctx;   /* one stream + scratch; one per thread   */

typedef struct kgpu_dict_info {
    uint64_t da_len;        /* double-array nodes (8 B each)                   */
    uint64_t n_morphs;      /* known morphs                                    */
    uint64_t n_unk_morphs;  /* unknown morphs                                  */
    uint64_t conn_rows;     /* right-id dimension                              */
    uint64_t conn_cols;     /* left-id dimension                               */
    uint64_t device_bytes;  /* HBM held by the dictionary                      */
    int32_t device;         /* HIP device ordinal                              */
    int32_t reserved;
} kgpu_dict_info;

/* Per-launch timing of the dominant kernel, collected with HIP events on the
 * ctx stream (bench.py roofline leg).  24 bytes, layout frozen: callers built against round 1 of this header
 * pass a buffer of exactly this size. */
typedef struct kgpu_profile {
    uint64_t launches;     /* tokenize kernel launches timed                  */
    double tokenize_ms;    /* sum over the timed batches of the whole tokenize launch chain (dominant kernel, long-sentence and
                              last-resort kernels, and whatever time those small launches wait for a slot on a busy chip) */
    double aux_ms;         /* sum of scan + compaction kernel durations         */
} kgpu_profile;

/* Routing counters, always on (they cost nothing: read from the batch's control block at
 * kgpu_ctx_sync).  A dictionary or text whose lattices outgrow the LDS-resident kernel shows up
 * here long before it shows up as a throughput cliff.  Read with kgpu_ctx_get_routing, which takes the
 * caller's sizeof(kgpu_routing): fields may be appended in later versions, never moved. */
typedef struct kgpu_routing {
    uint64_t batches;         /* batches completed                                            */
    uint64_t sentences;       /* sentences in them                                            */
    uint64_t deferred[4];     /* sentences handed from launch k of the chain to launch k+1
                                 ([0]: left the LDS-resident kernel for the windowed kernel;
                                 [1]: left that one for the general, HBM-scratch kernel)       */
    uint64_t redone[4];       /* ... of which only after the trie walk had been paid for
                                 (LDS reservation too small: the sentence was redone)         */
    uint64_t long_launches;   /* batches for which the windowed (long-sentence) kernel was launched */
    uint64_t arena_regrows;   /* batches rerun because the HBM scratch arena was too small    */
    double first_ms;          /* with KGPU_PROFILE_EVENTS: sum over the timed batches of the FIRST launch alone, the dominant
                                 kernel (k_tokenize_pool) -- the number rocprofv3's kernel stats report for it        */
    uint64_t small_calls;     /* kgpu_tokenize_batch calls served by the single-launch path                          */
    uint64_t small_fallbacks; /* ... that had to be redone on the general path (a sentence too long for LDS, or the
                                 in-kernel rendezvous timed out)                                                     */
    uint64_t window_reruns;   /* batches rerun with the HBM-lattice kernel because the windowed long-sentence kernel
                                 handed a sentence back                                                              */
    uint64_t tail_reruns;     /* batches whose long-sentence tail was launched afterwards (over the last work list only) because
                                 the tail of the launch chain had been left out (no recent batch needed it) and a sentence did need it */
    uint64_t combined_calls;  /* small kgpu_tokenize_batch calls that shared their launch with other threads' calls (the combiner) */
    uint64_t combined_launches; /* ... and the launches they shared                                                      */
} kgpu_routing;

/* The launch plan a context runs with (SURVEY.md 8d cfg 5: "LDS bytes / workgroup and achieved occupancy" as data). */
typedef struct kgpu_plan_info {
    uint32_t compute_units;
    uint32_t pool_lds_bytes;          /* LDS-resident kernel: bytes of the page pool one workgroup owns            */
    uint32_t pool_wavefronts;         /* ... independent wavefronts (= sentences in flight) sharing it             */
    uint32_t pool_workgroups_per_cu;  /* ... workgroups resident per CU (occupancy API)                            */
    uint32_t pool_max_pages;          /* ... pages of 64 a sentence may take before it is routed to the long path  */
    uint32_t long_lds_bytes;          /* unused since round 4 (rounds 2-3: a second long-sentence kernel; removed) -- always 0; the three fields keep their places */
    uint32_t long_workgroups_per_cu;
    uint32_t long_workgroups;
    uint32_t window_lds_bytes;        /* windowed kernel (everything the pool kernel routes away: ~150 characters and more, any length; dense
                                         lattices): LDS per single-wavefront workgroup, 0 = off                          */
    uint32_t window_workgroups_per_cu;/* ... resident per CU (occupancy API)                                       */
    uint32_t window_workgroups;       /* ... grid of one launch                                                    */
    uint32_t streams;                 /* HIP streams the dictionary's NULL-stream contexts share: 4 when the process has GPU_MAX_HW_QUEUES >= 5
                                         (the library sets it to 16 itself when it is loaded before the HIP runtime initialises and the variable
                                         is unset), else 3 -- and kgpu_last_error() then carries a warning after kgpu_dict_create.
                                         NOTE: that setenv(GPU_MAX_HW_QUEUES=16) is PROCESS-WIDE -- it changes the hardware-queue allocation of every
                                         other HIP user in the process (PyTorch, RCCL); KGPU_NO_PREINIT=1 in the environment, or setting the variable
                                         yourself, leaves it alone (the library then runs on three shared streams, long batches included)        */
    uint32_t long_streams;            /* further streams, one per context up to this many, for batches whose chain starts with the windowed kernel
                                         (long sentences: average length >= KGPU_WINDOW_FIRST bytes, default 1024): what the hardware queues leave
                                         -- 8 with 16 queues, 2 with 8, 0 (such batches stay on `streams`) with HIP's default 4                */
    uint32_t window_first_bytes;      /* ... that threshold (0 = every chain starts with the pool kernel)                                     */
    uint32_t reserved[2];
} kgpu_plan_info;
int kgpu_ctx_get_plan(kgpu_ctx *c, kgpu_plan_info *out, size_t out_size);

/* Work counters of one or more batches, counted on the device when
 * kgpu_ctx_set_profiling(ctx, KGPU_PROFILE_WORK) is on (the
 * algorithmic-byte formulas are written in these).  Slower: not for timed runs. */
typedef struct kgpu_work {
    uint64_t sentences;
    uint64_t B; /* input bytes                                              */
    uint64_t C; /* input chars                                              */
    uint64_t T; /* double-array byte steps attempted (incl. the failing one) */
    uint64_t N; /* lattice nodes excluding BOS                              */
    uint64_t E; /* Viterbi relaxations (target, predecessor) pairs          */
    uint64_t K; /* emitted tokens                                           */
} kgpu_work;

#define KGPU_PROFILE_OFF 0
#define KGPU_PROFILE_EVENTS 1 /* HIP events around the kernels            */
#define KGPU_PROFILE_WORK 2   /* device-side work counters (kgpu_work)    */
#define KGPU_PROFILE_SAMPLED 4 /* with EVENTS: time every 4th launch only  */
#define KGPU_PROFILE_NO_T 8    /* with WORK: leave kgpu_work.T at 0 -- the byte-level walk that counts the reference's trie steps runs
                                  beside the product's character-level walk and distorts the walk phase of kgpu_ctx_get_phase_cycles */

const char *kgpu_last_error(void);
int kgpu_device_count(void);

/* Tokenizer::new(dict) (src/tokenizer.rs:12-14): parse + validate the blobs,
 * upload once to HBM of `device`.  A dictionary on which the reference would
 * panic at tokenize time (morph id / connection index / invoke_list index out
 * of bounds: src/lattice.rs:54,182,195, connection.rs:13) is rejected here with
 * KGPU_ERR_BAD_DICT instead. */
int kgpu_dict_create(const kgpu_dict_blobs *blobs, int device, kgpu_dict **out);
/* Create-time validation is stricter than the reference's lazy panics, on purpose (a device kernel
 * cannot panic): a dictionary is rejected as a whole if ANY morph carries a negative context id, if
 * the largest (left, right) id pair of the dictionary indexes outside the matrix, if a duplicate
 * count exceeds 65535 or names a missing record, or if a category that occurs in char_category has
 * no invoke_list entry -- even when the offending entry could never be reached by a lattice. */
/* The handle may be destroyed while contexts made from it are alive: the tables and the shared
 * streams are released when the last such context is destroyed. */
void kgpu_dict_destroy(kgpu_dict *d);
int kgpu_dict_get_info(const kgpu_dict *d, kgpu_dict_info *out);

/* Tokenizer::tokenize(&self, &str) -> Vec<Token> (src/tokenizer.rs:16-45) for a
 * batch of n sentences in host memory: sentence i is
 * utf8[offsets[i] .. offsets[i+1]) (offsets has n+1 entries).  Tokens are
 * written densely in sentence order; tok_offsets (n+1 entries) delimits each
 * sentence's Vec<Token>.  A sentence whose EOS is unreachable yields zero
 * tokens, as in the reference (src/lattice.rs:144-153).  status may be NULL.
 * Thread-safe per dict (&self, src/tokenizer.rs:16): each call checks a ctx
 * out of an internal pool.  token_capacity >= total_chars + n always suffices. */
int kgpu_tokenize_batch(kgpu_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                        kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets,
                        uint8_t *status, uint64_t *n_tokens);

/* The same call over several devices of one node (reference src/tokenizer.rs:16 is &self, Send + Sync: a server shards its
 * sentences; BASELINE cfg 4: "sharded round-robin"): sentence i goes to dicts[i mod n_dicts] -- one dictionary handle per device,
 * made by kgpu_dict_create on that device (the same handle may appear more than once: its device then takes several shards) --
 * one host thread per entry drives its device's chunk pipeline, every device's compaction kernel writes its 8-byte records into
 * pinned host memory, and the records are expanded into `tokens` in the caller's ORIGINAL sentence order: the result is
 * byte-for-byte what kgpu_tokenize_batch gives on one device.  No data-path collective: the shards are independent.
 * EXPERIMENTAL (this entry point and kgpu_multi_* below): exercised with one device appearing several times; never yet run with handles on two physical
 * GPUs (tests/test_gpu_multi.py::test_one_handle_per_device does as soon as a box has two).  One difference from kgpu_tokenize_batch: a token that does not
 * fit the 8-byte record the shards write (more than 4095 characters or 262143 bytes: no real dictionary) fails the whole call with KGPU_ERR_INTERNAL, where
 * the single-device call redoes that chunk with 24-byte records.  The caller's current HIP device is restored before the call returns. */
int kgpu_tokenize_batch_multi(kgpu_dict *const *dicts, int n_dicts, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                              kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens);
/* ... with the records left as the devices produce them (round 6): `tokens8` receives the 8-byte kgpu_token8 records and first[2 i], first[2 i + 1] the
 * (position, start) of sentence i's first token, both in the caller's ORIGINAL sentence order; kgpu_expand_tokens(tokens8 + tok_offsets[a], tok_offsets + a,
 * first + 2 a, b - a, out) restores the 24-byte kgpu_token records of sentences [a, b) wherever and whenever the caller consumes them (or never: id, class,
 * lengths are all in the 8-byte record).  The 24-byte form's expansion writes ~1 KB of host memory per 40-character sentence on the calling side and binds
 * at about two devices' worth of records on 16 host CPUs; this form's merge moves a third of that (reference src/token.rs:10-18: a Token is id, class and
 * three positions -- the positions are running sums of the lengths).  Same sharding, same status bytes, same errors, same EXPERIMENTAL note. */
int kgpu_tokenize_batch_multi_compact(kgpu_dict *const *dicts, int n_dicts, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                                      kgpu_token8 *tokens8, uint64_t token_capacity, uint32_t *first, uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens);

/* Device-resident multi-device step with the result gather (the "trivial result gather over xGMI"): shard g -- n[g] sentences, packed
 * as kgpu_tokenize_device wants them -- is resident on dicts[g]'s device; its compaction kernel stores the 8-byte records (and the
 * per-sentence first position / start, token offsets and status bytes) straight into the ROOT device's memory through peer access
 * (root_* [g]: device pointers on dicts[0]'s device, one set per shard) -- the stores are the transfer, there is no copy node and no
 * collective.  `slot` in [0, slots): independent sets of contexts, so that several steps are in flight; kgpu_multi_sync(slot) waits
 * for that slot's shards and reports the token count of each. */
typedef struct kgpu_multi kgpu_multi;
int kgpu_multi_create(kgpu_dict *const *dicts, int n_dicts, int slots, kgpu_multi **out);
void kgpu_multi_destroy(kgpu_multi *m);
int kgpu_multi_tokenize_device(kgpu_multi *m, int slot, const uint8_t *const *d_utf8, const uint64_t *const *d_offsets, const uint64_t *n,
                               const uint64_t *total_bytes, kgpu_token8 *const *root_tokens8, const uint64_t *token_capacity,
                               uint32_t *const *root_first, uint64_t *const *root_tok_offsets, uint8_t *const *root_status);
int kgpu_multi_sync(kgpu_multi *m, int slot, uint64_t *n_tokens /* [n_dicts], may be NULL */);

/* Pinned host memory for the buffers of kgpu_tokenize_batch (optional): with it the host<->device copies of a
 * large call run as DMA and overlap the kernels of its other chunks; pageable buffers work, more slowly. */
void *kgpu_host_alloc(uint64_t bytes);
void kgpu_host_free(void *p);

/* Device-resident form (inputs already in HBM, outputs left in HBM, e.g. for
 * the RCCL gather).  All d_* pointers are device pointers on the dict's
 * device.  kgpu_tokenize_device only enqueues on the ctx stream;
 * kgpu_ctx_sync waits and reports the dense token count (or KGPU_ERR_CAPACITY). */
int kgpu_ctx_create(kgpu_dict *d, void *hip_stream /* NULL: one of the dictionary's shared streams (kgpu_plan_info.streams of them) */, kgpu_ctx **out);
/* One batch in flight per ctx; keep four or more contexts busy to fill the chip (eight for cfg 2's 4096-sentence batches).  Contexts may share a
 * stream (each waits on its own completion event); those created with NULL share kgpu_plan_info.streams per dictionary, and a batch of long
 * sentences is moved to one of kgpu_plan_info.long_streams for its duration.  With a caller-owned stream everything stays on that stream. */
void kgpu_ctx_destroy(kgpu_ctx *c);
int kgpu_tokenize_device(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n,
                         uint64_t total_bytes, kgpu_token *d_tokens, uint64_t token_capacity,
                         uint64_t *d_tok_offsets, uint8_t *d_status);
int kgpu_ctx_sync(kgpu_ctx *c, uint64_t *n_tokens);
/* Same batch, results as 8-byte records: d_tokens8 (token_capacity entries), d_first (2 n entries: position and start of
 * each sentence's first token, 0xFFFFFFFF twice for a sentence without tokens), d_tok_offsets, d_status.  The output
 * pointers may be device pointers of pinned, device-mapped host memory: the compaction kernel's stores are then the
 * device-to-host transfer.  A token that does not fit the packing (more than 4095 chars or 262143 bytes: no real
 * dictionary) makes kgpu_ctx_sync return KGPU_ERR_CAPACITY with *n_tokens = 0; use the 24-byte form for such a batch. */
int kgpu_tokenize_device_compact(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n,
                                 uint64_t total_bytes, kgpu_token8 *d_tokens8, uint64_t token_capacity,
                                 uint32_t *d_first, uint64_t *d_tok_offsets, uint8_t *d_status);
/* Host side: 8-byte records of n sentences -> the 24-byte records (running sums per sentence).  tok_offsets: n + 1
 * entries delimiting the sentences inside `in`; `out` receives tok_offsets[n] - tok_offsets[0] records. */
void kgpu_expand_tokens(const kgpu_token8 *in, const uint64_t *tok_offsets, const uint32_t *first, uint64_t n, kgpu_token *out);
int kgpu_ctx_set_profiling(kgpu_ctx *c, int mode /* KGPU_PROFILE_* bit mask */);
int kgpu_ctx_get_profile(kgpu_ctx *c, kgpu_profile *out, int reset);
/* Copies min(out_size, sizeof(kgpu_routing)) bytes: a caller built against an older, shorter kgpu_routing stays valid. */
int kgpu_ctx_get_routing(kgpu_ctx *c, kgpu_routing *out, size_t out_size, int reset);
/* The same counters summed over the dictionary's pooled contexts -- the ones kgpu_tokenize_batch and kgpu_lattice_dump check out per call (idle
 * ones only: a context serving a call right now is counted once it is back).  This is where small_calls / combined_calls show. */
int kgpu_dict_get_routing(kgpu_dict *d, kgpu_routing *out, size_t out_size, int reset);
/* Measurement only (bench.py's per-stage roofline): every sentence of the following batches stops
 * after the given stage of the fused kernel, yields zero tokens and status KGPU_SENT_TRUNCATED.
 * 0 = off (normal operation).  Stages: 5 = lattice built (SURVEY.md 8d Stage A: load, decode, trie
 * walk, numbering, node emission), 7 = Viterbi sweep done (Stage B: connection-cost gather + sweep);
 * the remainder is Stage C (backtrace + token records). */
#define KGPU_STAGE_ALL 0
#define KGPU_STAGE_LATTICE 5
#define KGPU_STAGE_GATHER 6
#define KGPU_STAGE_VITERBI 7
int kgpu_ctx_set_ablation(kgpu_ctx *c, int stop_after_stage);
int kgpu_ctx_get_work(kgpu_ctx *c, kgpu_work *out, int reset);
/* Sum over sentences of shader-clock cycles spent per phase of the LDS-resident
 * kernel (KGPU_PROFILE_WORK runs only): load, decode, walk, scan, emit, gather,
 * sweep, backtrace+tokens, [8] = sentences counted, [9] spare. */
int kgpu_ctx_get_phase_cycles(kgpu_ctx *c, uint64_t out[10], int reset);

/* Lattice (reference src/lattice.rs:6-10: pub nodes, pub edges) of ONE sentence, read back from the device after
 * Lattice::build + the forward pass of Lattice::viterbi -- the debugging aid behind the reference's `kanpyo graphviz`
 * (src/graphviz.rs:30-163, src/bin/kanpyo.rs:127-148).  Nodes are in the reference's insertion order (BOS = 0, EOS last);
 * edges[e] = the nodes ENDING at char position e, ascending, as offsets into edge_nodes (n_positions + 1 entries,
 * n_positions = chars + 2).  Context ids are the dictionary's own (not the device's frequency-ranked ones).
 * dp / pre are the Viterbi state of src/lattice.rs:118-141: dp = 1 << 30 where unreached, BOS has dp 0 ("None") and
 * pre -1.  Runs the HBM-scratch kernel alone (any sentence length); not a fast path.  Free with kgpu_lattice_free. */
typedef struct kgpu_lattice_node {
    int32_t id;          /* Node::id(): KeywordID, 0 for BOS / EOS              (src/lattice/node.rs:27-32) */
    uint32_t cls;        /* KGPU_CLASS_DUMMY / KNOWN / UNKNOWN                                              */
    uint32_t byte_pos;   /* byte offset of the surface in the sentence                                      */
    uint32_t char_pos;   /* char index where the node starts                                                */
    uint32_t end_char;   /* char index where it ends (EOS: char_pos; BOS: 0)                                */
    uint32_t byte_len;   /* surface length in bytes                                                         */
    int16_t left_id, right_id, cost, reserved; /* Morph (kanpyo-dict/src/morph.rs:7-11)                     */
    int32_t dp;          /* best total cost up to and including this node                                   */
    int32_t pre;         /* best predecessor node index, -1 = None                                          */
} kgpu_lattice_node;
typedef struct kgpu_lattice {
    uint64_t n_nodes, n_positions;
    kgpu_lattice_node *nodes;
    uint32_t *edge_offsets; /* n_positions + 1 */
    uint32_t *edge_nodes;   /* n_nodes         */
} kgpu_lattice;
int kgpu_lattice_dump(kgpu_dict *d, const uint8_t *utf8, uint64_t len, kgpu_lattice *out);
void kgpu_lattice_free(kgpu_lattice *l);

/* IndexTable::build + write_dict (kanpyo-dict/src/index.rs:16-38,75-84 over
 * trie/da.rs:22-131,191-217): sorted keywords (duplicates adjacent) -> the
 * index.dict blob, byte-identical to the reference's first-fit packing.  Host
 * only.  Free the blob with kgpu_free. */
int kgpu_index_build(const uint8_t *keys, const uint64_t *key_offsets, uint64_t n,
                     uint8_t **blob, size_t *blob_len);
void kgpu_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
