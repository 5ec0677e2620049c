/* tests/c_abi/consumer.c -- a compiled C consumer of include/kanpyo_gpu.h: the closest thing to the Rust shim of
 * INTEGRATION.md this image can build (no rustc).  C99, includes ONLY the public header and links libkanpyo_gpu.so.
 *
 *   consumer <dir>
 *
 * <dir> holds the dictionary as the reference serialises it (DictReadWrite::write_dict, kanpyo-dict/src/dict.rs:13-18) minus
 * the index: keywords.bin / keywords.off (the sorted keyword list: the consumer builds index.dict itself through
 * kgpu_index_build, as IndexTable::build does, kanpyo-dict/src/index.rs:16-38), connection.dict, morph.dict, unk.dict,
 * char_category.bin, invoke.bin, group.bin, and sentences.bin / sentences.off (uint64 offsets, n + 1 of them).
 * It calls kgpu_dict_create / kgpu_tokenize_batch (what Tokenizer::new / Tokenizer::tokenize bind, reference
 * src/tokenizer.rs:12-45) / a device context's routing counters / kgpu_lattice_dump and prints every record:
 *   S <i> <status> <n_tokens>          one per sentence
 *   T <id> <cls> <position> <start> <end> <byte_len>
 *   R <batches> <sentences>            routing counters of a device-resident call over the same sentences
 *   L <n_nodes> <n_positions>          lattice of sentence 0
 *   N <id> <cls> <byte_pos> <char_pos> <end_char> <byte_len> <left> <right> <cost> <dp> <pre>
 * tests/test_c_abi_gpu.py compares the output with tests/golden/fixture_tokens.json. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kanpyo_gpu.h"

static uint8_t *slurp(const char *dir, const char *name, size_t *len) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *p = (uint8_t *)malloc((size_t)n + 1);
    if (n && fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    *len = (size_t)n;
    return p;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != KGPU_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, kgpu_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: consumer <dir>\n"); return 2; }
    const char *dir = argv[1];
    size_t kw_len, kwo_len, s_len, so_len;
    uint8_t *kw = slurp(dir, "keywords.bin", &kw_len);
    uint64_t *kwo = (uint64_t *)slurp(dir, "keywords.off", &kwo_len);
    uint8_t *sent = slurp(dir, "sentences.bin", &s_len);
    uint64_t *soff = (uint64_t *)slurp(dir, "sentences.off", &so_len);
    const uint64_t n_kw = kwo_len / 8 - 1, n = so_len / 8 - 1;

    kgpu_dict_blobs b;
    memset(&b, 0, sizeof b);
    uint8_t *index_blob = NULL;
    size_t index_len = 0;
    CHECK(kgpu_index_build(kw, kwo, n_kw, &index_blob, &index_len));
    b.index_dict = index_blob; b.index_len = index_len;
    b.connection_dict = slurp(dir, "connection.dict", &b.connection_len);
    b.morph_dict = slurp(dir, "morph.dict", &b.morph_len);
    b.unk_dict = slurp(dir, "unk.dict", &b.unk_len);
    b.char_category = slurp(dir, "char_category.bin", &b.char_category_len);
    b.invoke_list = slurp(dir, "invoke.bin", &b.invoke_len);
    b.group_list = slurp(dir, "group.bin", &b.group_len);

    if (kgpu_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
    kgpu_dict *d = NULL;
    CHECK(kgpu_dict_create(&b, 0, &d));
    kgpu_dict_info info;
    CHECK(kgpu_dict_get_info(d, &info));
    printf("D %llu %llu %llu %llu\n", (unsigned long long)info.n_morphs, (unsigned long long)info.n_unk_morphs,
           (unsigned long long)info.conn_rows, (unsigned long long)info.conn_cols);

    /* Tokenizer::tokenize for every sentence in one call (host buffers) */
    const uint64_t cap = soff[n] - soff[0] + n + 1;
    kgpu_token *tok = (kgpu_token *)malloc((size_t)cap * sizeof *tok);
    uint64_t *toff = (uint64_t *)malloc((size_t)(n + 1) * 8);
    uint8_t *status = (uint8_t *)calloc((size_t)n + 1, 1);
    uint64_t got = 0;
    CHECK(kgpu_tokenize_batch(d, sent, soff, n, tok, cap, toff, status, &got));
    for (uint64_t i = 0; i < n; ++i) {
        printf("S %llu %u %llu\n", (unsigned long long)i, (unsigned)status[i], (unsigned long long)(toff[i + 1] - toff[i]));
        for (uint64_t k = toff[i]; k < toff[i + 1]; ++k)
            printf("T %d %u %u %u %u %u\n", tok[k].id, tok[k].cls, tok[k].position, tok[k].start, tok[k].end, tok[k].byte_len);
    }
    /* ... and one sentence per call, the reference's own call shape (src/bin/kanpyo.rs:106-126): same records */
    for (uint64_t i = 0; i < n; ++i) {
        kgpu_token one[64];
        uint64_t o2[2], g2 = 0;
        uint8_t st = 0;
        const uint64_t off2[2] = {soff[i], soff[i + 1]};
        CHECK(kgpu_tokenize_batch(d, sent, off2, 1, one, 64, o2, &st, &g2));
        if (g2 != toff[i + 1] - toff[i] || st != status[i] || memcmp(one, tok + toff[i], (size_t)g2 * sizeof *one) != 0) { fprintf(stderr, "single call differs at %llu\n", (unsigned long long)i); return 4; }
    }

    /* the 8-byte record form + routing counters through a device context: this path needs device memory, which plain C
     * cannot allocate without the HIP headers -- pinned host memory from kgpu_host_alloc is device-accessible, so it serves */
    {
        kgpu_ctx *c = NULL;
        CHECK(kgpu_ctx_create(d, NULL, &c));
        const size_t nb = (size_t)(soff[n] - soff[0]);
        uint8_t *p_utf8 = (uint8_t *)kgpu_host_alloc(nb + 16);
        uint64_t *p_off = (uint64_t *)kgpu_host_alloc((n + 1) * 8);
        kgpu_token8 *p_t8 = (kgpu_token8 *)kgpu_host_alloc(cap * 8);
        uint32_t *p_first = (uint32_t *)kgpu_host_alloc((n + 1) * 8);
        uint64_t *p_toff = (uint64_t *)kgpu_host_alloc((n + 1) * 8);
        uint8_t *p_st = (uint8_t *)kgpu_host_alloc(n + 16);
        if (!p_utf8 || !p_off || !p_t8 || !p_first || !p_toff || !p_st) { fprintf(stderr, "kgpu_host_alloc: %s\n", kgpu_last_error()); return 1; }
        memcpy(p_utf8, sent + soff[0], nb);
        for (uint64_t i = 0; i <= n; ++i) p_off[i] = soff[i] - soff[0];
        CHECK(kgpu_tokenize_device_compact(c, p_utf8, p_off, n, nb, p_t8, cap, p_first, p_toff, p_st));
        uint64_t g8 = 0;
        CHECK(kgpu_ctx_sync(c, &g8));
        kgpu_token *back = (kgpu_token *)malloc((size_t)(g8 + 1) * sizeof *back);
        kgpu_expand_tokens(p_t8, p_toff, p_first, n, back);
        if (g8 != got || memcmp(back, tok, (size_t)got * sizeof *tok) != 0 || memcmp(p_toff, toff, (size_t)(n + 1) * 8) != 0) { fprintf(stderr, "8-byte records differ\n"); return 5; }
        kgpu_routing rt;
        memset(&rt, 0, sizeof rt);
        CHECK(kgpu_ctx_get_routing(c, &rt, sizeof rt, 0));
        printf("R %llu %llu\n", (unsigned long long)rt.batches, (unsigned long long)rt.sentences);
        kgpu_plan_info pl;
        CHECK(kgpu_ctx_get_plan(c, &pl, sizeof pl));
        printf("P %u %u\n", pl.compute_units, pl.pool_wavefronts);
        kgpu_ctx_destroy(c);
        kgpu_host_free(p_utf8); kgpu_host_free(p_off); kgpu_host_free(p_t8); kgpu_host_free(p_first); kgpu_host_free(p_toff); kgpu_host_free(p_st);
        free(back);
    }

    /* Lattice of sentence 0 (reference src/lattice.rs:6-10, `kanpyo graphviz`) */
    {
        kgpu_lattice lat;
        CHECK(kgpu_lattice_dump(d, sent + soff[0], soff[1] - soff[0], &lat));
        printf("L %llu %llu\n", (unsigned long long)lat.n_nodes, (unsigned long long)lat.n_positions);
        for (uint64_t t = 0; t < lat.n_nodes; ++t) {
            const kgpu_lattice_node *nd = &lat.nodes[t];
            printf("N %d %u %u %u %u %u %d %d %d %d %d\n", nd->id, nd->cls, nd->byte_pos, nd->char_pos, nd->end_char, nd->byte_len,
                   nd->left_id, nd->right_id, nd->cost, nd->dp, nd->pre);
        }
        kgpu_lattice_free(&lat);
    }
    kgpu_dict_destroy(d);
    kgpu_free(index_blob);
    return 0;
}
