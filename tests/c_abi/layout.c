/* tests/c_abi/layout.c -- prints sizeof / offsetof of every struct include/kanpyo_gpu.h declares, as
 * "struct field offset size" lines (field "-" = the whole struct).  C99, includes only the public header.
 * tests/test_c_abi_cpu.py compiles it with gcc and compares the output with the ctypes mirror (kanpyo_amd/_lib.py,
 * kanpyo_amd/tokenizer.py): the boundary a Rust shim binds (reference src/tokenizer.rs:7-45, src/token.rs:10-18) is
 * checked against the header itself, not against its own mirror. */
#include <stddef.h>
#include <stdio.h>

#include "kanpyo_gpu.h"

#define S(T) printf("%s - 0 %zu\n", #T, sizeof(T))
#define F(T, f) printf("%s %s %zu %zu\n", #T, #f, offsetof(T, f), sizeof(((T *)0)->f))

int main(void) {
    S(kgpu_token); F(kgpu_token, id); F(kgpu_token, cls); F(kgpu_token, position); F(kgpu_token, start); F(kgpu_token, end); F(kgpu_token, byte_len);
    S(kgpu_token8); F(kgpu_token8, id); F(kgpu_token8, packed);
    S(kgpu_dict_blobs);
    F(kgpu_dict_blobs, index_dict); F(kgpu_dict_blobs, index_len); F(kgpu_dict_blobs, connection_dict); F(kgpu_dict_blobs, connection_len);
    F(kgpu_dict_blobs, morph_dict); F(kgpu_dict_blobs, morph_len); F(kgpu_dict_blobs, unk_dict); F(kgpu_dict_blobs, unk_len);
    F(kgpu_dict_blobs, char_category); F(kgpu_dict_blobs, char_category_len); F(kgpu_dict_blobs, invoke_list); F(kgpu_dict_blobs, invoke_len);
    F(kgpu_dict_blobs, group_list); F(kgpu_dict_blobs, group_len);
    S(kgpu_dict_info);
    F(kgpu_dict_info, da_len); F(kgpu_dict_info, n_morphs); F(kgpu_dict_info, n_unk_morphs); F(kgpu_dict_info, conn_rows); F(kgpu_dict_info, conn_cols);
    F(kgpu_dict_info, device_bytes); F(kgpu_dict_info, device); F(kgpu_dict_info, reserved);
    S(kgpu_profile); F(kgpu_profile, launches); F(kgpu_profile, tokenize_ms); F(kgpu_profile, aux_ms);
    S(kgpu_routing);
    F(kgpu_routing, batches); F(kgpu_routing, sentences); F(kgpu_routing, deferred); F(kgpu_routing, redone); F(kgpu_routing, long_launches);
    F(kgpu_routing, arena_regrows); F(kgpu_routing, first_ms); F(kgpu_routing, small_calls); F(kgpu_routing, small_fallbacks);
    F(kgpu_routing, window_reruns); F(kgpu_routing, tail_reruns); F(kgpu_routing, combined_calls); F(kgpu_routing, combined_launches);
    S(kgpu_plan_info);
    F(kgpu_plan_info, compute_units); F(kgpu_plan_info, pool_lds_bytes); F(kgpu_plan_info, pool_wavefronts); F(kgpu_plan_info, pool_workgroups_per_cu);
    F(kgpu_plan_info, pool_max_pages); F(kgpu_plan_info, long_lds_bytes); F(kgpu_plan_info, long_workgroups_per_cu); F(kgpu_plan_info, long_workgroups);
    F(kgpu_plan_info, window_lds_bytes); F(kgpu_plan_info, window_workgroups_per_cu); F(kgpu_plan_info, window_workgroups); F(kgpu_plan_info, streams); F(kgpu_plan_info, long_streams); F(kgpu_plan_info, window_first_bytes); F(kgpu_plan_info, reserved);
    S(kgpu_work);
    F(kgpu_work, sentences); F(kgpu_work, B); F(kgpu_work, C); F(kgpu_work, T); F(kgpu_work, N); F(kgpu_work, E); F(kgpu_work, K);
    S(kgpu_lattice_node);
    F(kgpu_lattice_node, id); F(kgpu_lattice_node, cls); F(kgpu_lattice_node, byte_pos); F(kgpu_lattice_node, char_pos); F(kgpu_lattice_node, end_char);
    F(kgpu_lattice_node, byte_len); F(kgpu_lattice_node, left_id); F(kgpu_lattice_node, right_id); F(kgpu_lattice_node, cost); F(kgpu_lattice_node, reserved);
    F(kgpu_lattice_node, dp); F(kgpu_lattice_node, pre);
    S(kgpu_lattice);
    F(kgpu_lattice, n_nodes); F(kgpu_lattice, n_positions); F(kgpu_lattice, nodes); F(kgpu_lattice, edge_offsets); F(kgpu_lattice, edge_nodes);
    return 0;
}
