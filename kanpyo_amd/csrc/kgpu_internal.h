// kanpyo_amd/csrc/kgpu_internal.h -- shared between the host runtime
// (kgpu_api.cpp, kgpu_index_build.cpp) and the HIP kernels (kgpu_kernels.hip)
// of libkanpyo_gpu.so.  Not part of the public ABI.  Plain structs only, so it
// compiles as host C++ and as HIP.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/kanpyo_gpu.h"

namespace kgpu {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// ---- HBM-resident dictionary (uploaded once by kgpu_dict_create) ----------
struct alignas(8) DaNode {  // trie/da.rs:13-17: one {base, check} pair = one 64-bit load
    int32_t base, check;
};
struct alignas(8) Morph8 {  // morph.rs:7-11 padded to 8 bytes
    int16_t left, right, cost;
    uint16_t dup;  // IndexTable.dup[id] (index.rs:12,47) when this record is the
                   // first of its surface, else 0; rides in the padding
};
struct alignas(16) CatInfo {  // one per category byte value
    uint32_t flags;  // bit0 invoke_list[cat], bit1 group_list[cat], bit2 unk entry present
    int32_t unk_first;   // UnkDict.char_category_to_morph_id[cat].0 (unk_dict.rs:15)
    uint32_t unk_count;  //                                        .1
    uint32_t pad;
};
enum : uint32_t { CAT_INVOKE = 1u, CAT_GROUP = 2u, CAT_HAS_UNK = 4u };
// The character-level copy of the trie (kgpu_chartrie.cpp).  A node: one 16-byte load tells a walk whether the child exists (check), where its
// children are (base) and whether a key ends on it (leaf < 0: the byte-level leaf's base, i.e. -(id | dup << 21) or -id) -- no terminator probe.
struct alignas(16) CtNode { int32_t base, check, leaf, pad; };
// ... and what a walk needs to know about the character it starts with, in one 16-byte load.
struct alignas(16) CharRec {
    int32_t base, slot;   // the root's child for this character in the char-level array: its base and its slot (0: no key starts with it)
    uint16_t code;        // the character's code in the char-level array (0xFFFF: in no key)
    uint8_t cat, pad;     // char_category_def.rs:33-38, resolved (table[cp] if in range else table[0])
    int32_t leaf;         // < 0: this character alone is a key (CtNode::leaf of the root's child)
};
struct CharTrie {         // host side, before the upload
    std::vector<CtNode> da;
    std::vector<CharRec> rec;                // [65536]
    std::vector<uint32_t> nb_cp, nb_code;    // characters >= U+FFFF that occur in keys, ascending, and their codes
    uint32_t n_codes = 0;
};
bool build_char_trie(const std::vector<DaNode> &da, const uint8_t *cat, size_t cat_len, CharTrie &out);  // false: walk the bytes

struct DictView {
    const DaNode *da;        uint32_t da_len;
    uint32_t leaf_dup;       // 1: a leaf's base is -(id | dup << 21), dup = IndexTable.dup[id] (1023: look it up in the morph record):
                             // the device copy of the double array is re-encoded at create time so that the walk learns a surface's
                             // record count (index.rs:46-51) from the terminator node it has to load anyway
    const DaNode *first;     // [65536] per BMP code point: {node, base[node]} after walking its UTF-8 bytes from
                             // the root, or {0, byte steps attempted before the walk failed}
    const Morph8 *morph;     uint32_t n_morph;
    const Morph8 *unk_morph; uint32_t n_unk_morph;   // == morph + n_morph: one table (record of node sid: morph[sid > 0 ? sid - 1 : n_morph - 1 - sid])
    const int16_t *conn;     uint32_t conn_rows;  // element (right,left) at left*rows+right (ids frequency-ranked, see kgpu_api.cpp)
    uint32_t bos_right, eos_left;                  // the ranked ids of context id 0 (BOS/EOS Morph(0,0,0))
    const uint8_t *cat;      uint32_t cat_len;    // char_category_def.rs:17,33-38
    const CatInfo *cinfo;                          // 256 entries
    // Character-level double array (kgpu_chartrie.cpp; nullptr: the dictionary is walked byte by byte): one dependent load per character
    // instead of one per byte.  crec: per BMP code point; nb_cp / nb_code: the few characters >= U+FFFF that occur in keys.
    const CtNode *da2;       uint32_t da2_len; uint32_t n_nb;
    const CharRec *crec;
    const uint32_t *nb_cp;   const uint32_t *nb_code;
};

// ---- per-ctx control block in device memory (zeroed before every batch) ---
struct Control {
    unsigned int ovf_count[4];        // sentences handed from launch k to launch k+1
    unsigned int late_count[4];       // ... of which only after the trie walk had been paid for
    unsigned long long arena_cursor;  // bump allocator over the scratch arena (bytes)
    unsigned int arena_overflow;      // a slab request did not fit
    unsigned int pad0;
    unsigned long long n_tokens;      // dense token count (written by the scan kernel)
    unsigned long long work[7];       // kgpu_work, only when BatchArgs::count_work
    unsigned long long phase[10];     // shader-clock cycles per phase of the LDS kernel (count_work only)
    unsigned int waves_done;          // single-launch small calls: wavefronts through with their sentence ...
    unsigned int waves_copied;        // ... and through with moving its tokens to the caller's (pinned) buffers
    unsigned int small_flag;          // the call's sequence number, stored LAST into the host copy: the host polls it
    unsigned int pack_overflow;       // compact records: a token did not fit kgpu_token8 (chars > 4095 or bytes > 262143): the host falls back to 24-byte records
    unsigned int window_fail;         // the windowed long-sentence kernel met a sentence it cannot hold AND had no list to hand it on to: the host reruns the batch with the HBM-lattice kernel
    unsigned int small_abort;         // ... a wavefront gave up waiting at the rendezvous: the host redoes the call on the general path
    unsigned int win_ticket;          // the windowed kernel's ordinary form: next entry of its work list (its workgroups claim sentences one by one)
    unsigned int pad3;
    unsigned long long dump[8];       // kgpu_lattice_dump: arena offsets of the sentence's two slabs, B, C, N, 1 = valid, dp of EOS
};

struct BatchArgs {
    const uint8_t *utf8;          // concatenated sentences
    const uint64_t *offsets;      // n+1
    uint64_t n;
    Control *ctl;
    uint8_t *arena;  uint64_t arena_bytes;
    kgpu_token *stage;            // staging tokens: sentence s owns slots [off[s]-off[0]+s, +B_s+1)
    uint32_t *tok_count;          // n
    uint8_t *status;              // n
    kgpu_token *out;  uint64_t out_cap;      // dense output
    uint64_t *tok_offsets;        // n+1
    uint32_t count_work;          // accumulate kgpu_work into ctl->work (slow; off in timed runs)
    uint32_t *ovf[4];             // n entries each: work lists of launches 1.. (filled by the launch before)
    uint32_t est_q8;              // expected LDS bytes per input byte (x256): reservation size and length routing
    uint32_t dump_lattice;        // general kernel: leave the slab offsets of the (single) sentence in ctl->dump
    Control *fused_host;          // non-null: single-launch small call -- the pool kernel also scans, compacts into the (pinned,
    uint32_t fused_seq;           // device-mapped) output and publishes the control block with this sequence number
    unsigned long long *stat_slots;  // profiling runs: STAT_SLOTS x STAT_WORDS counters, one slot per wavefront of the pool launch
    // compact result records (host-buffer path, kgpu_tokenize_device_compact): when out8 is set the compaction kernel writes 8-byte
    // kgpu_token8 records (and per sentence the first token's position / start) instead of 24-byte kgpu_token records;
    // out8 / first8 may be device-mapped pinned host memory: the kernel's stores then ARE the device-to-host transfer
    kgpu_token8 *out8;
    uint32_t *first8;                // [2 n]: position, start of sentence s's first token (0xFFFFFFFF twice: no tokens)
    uint8_t *status8;                // optional (host path): the compaction kernel mirrors status[] there (mapped host memory)
    uint64_t *toff8;                 // optional (host path): ... and tok_offsets[] (n + 1), so that it reads its offsets from HBM, not back over PCIe
};                                   // (added to by its owner, summed on the host: hot atomics on a few words would distort the run)
constexpr uint32_t STAT_SLOTS = 16384, STAT_WORDS = 32;  // words 0..6: Control::work, 16..25: Control::phase

// Launch plan of one batch: the LDS page-pool kernel (kgpu_pool.hip) once or twice -- W independent
// wavefronts per workgroup share pool_bytes of LDS, each sentence takes what it needs -- then the windowed kernel
// (kgpu_window.hip: bounded LDS whatever the length) for whatever fits no pool, then the general kernel, whose lattices
// live in HBM scratch, as the last resort.  A sentence that a launch cannot serve is pushed onto the next launch's work list.
struct LaunchPlan {
    int n_pools;
    uint32_t pool_bytes[2];
    uint32_t pool_waves[2];
    uint32_t pool_max_pages[2];  // of 64: larger reservations are routed to the next launch
    bool pool_limit_auto;        // the shipped plan (no KGPU_POOL): the runtime may pick the pool shape per batch (kgpu_api.cpp: enqueue)
    uint32_t alt_pool_bytes, alt_pool_waves;   // ... the shape for chains that hold a windowed launch: smaller workgroups (20 KB, two wavefronts) find their LDS
    int alt_pool_workgroups;                   // sooner on a chip full of 10 KB single-wavefront workgroups (0: not available)
    int pool_workgroups[2];   // persistent grid per pool launch
    int general_workgroups;
    uint32_t window_lds_bytes;  // > 0: the windowed kernel (kgpu_window.hip) behind the pools; 0: the general kernel serves what they route away
    int window_workgroups;
    int window_team;             // wavefronts per sentence of the windowed kernel's team form (KGPU_WINDOW_TEAM_SIZE, default 2)
    int window_team_workgroups;  // resident workgroups of that form on the whole chip (0: not available)
    int window_team_mode;        // KGPU_WINDOW_TEAM: 0 never, 2 whenever the chain starts with the windowed kernel, -1 (default) by the load
    uint32_t window_first_bytes; // KGPU_WINDOW_FIRST: a batch averaging this many bytes per sentence or more gets no pool launch in front (default 1024; 0 = never)
};

// Launchers (kgpu_kernels.hip).  `stream` is a hipStream_t.
// n_pools_now <= plan.n_pools: how many of the pool launches to issue for this batch (the chain
// stays complete without the later ones: their work falls through to the next launch).
int launch_tokenize(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, int n_pools_now,
                    uint32_t stop_after /* kgpu_ctx_set_ablation; 0 = run everything */, void *stream,
                    void *event_after_first /* hipEvent_t recorded behind the first (dominant) launch, or null */,
                    bool window_now /* the windowed kernel behind the pools (plan.window_lds_bytes) */,
                    bool tail_now /* false: nothing behind a chain that has a work list (the host launches what is missing if a sentence needed it) */,
                    bool team_now = false /* a chain without pool launches: the windowed kernel's two-wavefronts-per-sentence form first, its ordinary form behind it */,
                    int window_grid = 0 /* > 0: workgroups of the windowed launch behind the pools (the host's estimate of its work list; any grid is correct, the list is strided) */);
int launch_tail_only(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, int list_index, bool window_was_in_chain, void *stream);
int window_workgroups_per_cu(uint32_t lds_bytes);
int window_team_workgroups_per_cu(uint32_t lds_bytes, int team);
int launch_general_only(const DictView &d, const BatchArgs &a, void *stream);
int launch_small_call(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, void *stream);  // pool kernel alone, one sentence per wavefront  // kgpu_lattice_dump: HBM-scratch kernel alone
int launch_scan_compact(const BatchArgs &a, Control *host_ctl, void *stream, bool small_workgroups = false, bool small_scan_only = false /* measurement */);  // host_ctl: device pointer of the pinned result block
LaunchPlan default_launch_plan(int device);
int pool_workgroups_per_cu(uint32_t pool_bytes, uint32_t waves);

}  // namespace kgpu
