"""ctypes binding of libelf_amd.so (include/elf_amd.h). There is NO fallback: if the HIP library is
missing or fails to load, importing the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libelf_amd.so")
ZOBRIST_BIN = os.path.join(HERE, "data", "zobrist21.bin")

_lib = None

# the sources every device kernel of libelf_amd.so is compiled from: measurements that are tied to the kernels' instruction
# streams (profiles/pmc_issue.json, pmc_traffic.json) record this hash, and bench.py only prices against them while it matches
KERNEL_SOURCES = ("go_board.cuh", "mcts.cuh", "train.cuh", "stl_emul.h", "engine_host.h", "elf_amd.hip", "mcts_capi.hip", "train_capi.hip",
                  "net_epilogue.hip", "Makefile")


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]

# name -> (restype, argtypes); mirrors include/elf_amd.h one to one
_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "elfgo_create": (_i, [_i, _i, _i, _vp, C.POINTER(_vp)]),
    "elfgo_destroy": (_i, [_vp]),
    "elfgo_board_size": (_i, [_vp]),
    "elfgo_capacity": (_i, [_vp]),
    "elfgo_slot_bytes": (_sz, [_vp]),
    "elfgo_sync": (_i, [_vp, _vp]),
    "elfgo_reset": (_i, [_vp, _vp, _i, _vp]),
    "elfgo_copy": (_i, [_vp, _vp, _vp, _i, _vp]),
    "elfgo_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "elfgo_legal_mask": (_i, [_vp, _vp, _i, _vp, _vp]),
    "elfgo_extract_agz": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp]),
    "elfgo_extract_agz_fmt": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i, _vp]),
    "elfgo_evaluate": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "elfgo_info": (_i, [_vp, _vp, _i, _vp, _vp]),
    "elfgo_export_board": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "elfgo_playout": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "elfmcts_create": (_i, [_vp, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "elfmcts_destroy": (_i, [_vp]),
    "elfmcts_set_options": (_i, [_vp, _vp]),
    "elfmcts_set_feature_format": (_i, [_vp, _i]),
    "elfmcts_get_feature_format": (_i, [_vp, C.POINTER(_i)]),
    "elfmcts_max_rollouts_per_step": (_i, []),
    "elfmcts_set_game_mask": (_i, [_vp, _vp]),
    "elfmcts_set_required_versions": (_i, [_vp, _vp]),
    "elfmcts_num_games": (_i, [_vp]),
    "elfmcts_edge_stride": (_i, [_vp]),
    "elfmcts_node_bytes": (_sz, [_vp]),
    "elfmcts_tree_bytes_per_game": (_sz, [_i, _i]),
    "elfmcts_tree_bytes_per_game2": (_sz, [_i, _i, _i, _i, _i]),
    "elfmcts_pool_info": (_i, [_vp, _vp, _i]),
    "elfmcts_count_live": (_i, [_vp, _vp]),
    "elfmcts_num_threads": (_i, [_vp]),
    "elfmcts_thread_draws": (_i, [_vp, _vp, _vp]),
    "elfmcts_clear": (_i, [_vp, _vp, _i, _vp]),
    "elfmcts_set_root": (_i, [_vp, _vp, _vp]),
    "elfmcts_set_d4": (_i, [_vp, _vp, _vp]),
    "elfmcts_dirichlet": (_i, [_vp, _vp, _vp, _f, _vp]),
    "elfmcts_select": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "elfmcts_expand": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "elfmcts_root": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "elfmcts_advance": (_i, [_vp, _vp, _vp]),
    "elfmcts_validate": (_i, [_vp, _vp]),
    "elfmcts_node_visits": (_i, [_vp, _vp]),
    "elfsp_create": (_i, [_vp, _i, _vp, C.POINTER(_vp)]),
    "elfsp_destroy": (_i, [_vp]),
    "elfsp_engine": (_vp, [_vp]),
    "elfsp_mcts": (_vp, [_vp]),
    "elfsp_ts_requests_deferred": (_i64, [_vp]),
    "elfsp_ts_games_deferred": (_i, [_vp]),
    "elfsp_max_rows": (_i, [_vp]),
    "elfsp_max_rows_actor": (_i, [_vp, _i]),
    "elfsp_mcts_actor": (_vp, [_vp, _i]),
    "elfsp_begin_step2": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "elfsp_end_step2": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "elfsp_last_rows2": (_i, [_vp, _vp]),
    "elfsp_set_request2": (_i, [_vp, _vp]),
    "elfsp_set_request3": (_i, [_vp, _vp, _vp]),
    "elfrec_sgf_parse": (_i, [_i, C.c_char_p, _vp, _vp, _i, _vp]),
    "elfrec_game_sgf": (_i64, [_vp, _vp, _i, _vp, _i, _f, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, _sz]),
    "elfrec_record_to_sgf": (_i64, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, _sz, C.c_char_p, _sz]),
    "elfrec_client_create": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "elfrec_client_destroy": (_i, [_vp]),
    "elfrec_client_feed": (_i, [_vp, C.c_char_p]),
    "elfrec_client_update_state": (_i, [_vp, _vp]),
    "elfrec_client_size": (_i, [_vp]),
    "elfrec_client_dump_and_clear": (_i64, [_vp, C.c_char_p, _sz]),
    "elfrec_parse_request_seq": (_i, [C.c_char_p, _vp, C.POINTER(_i64), _vp]),
    "elfrec_request_seq_to_json": (_i64, [_vp, _vp, _i64, C.c_char_p, _sz]),
    "elfsp_thread_states": (_i, [_vp, _vp, _i]),
    "elfrq_create": (_i, [_i, _i, _i, C.c_uint32, C.POINTER(_vp)]),
    "elfrq_destroy": (_i, [_vp]),
    "elfrq_insert": (_i, [_vp, C.c_int32, C.c_int32, _i, C.POINTER(C.c_int32)]),
    "elfrq_sizes": (_i, [_vp, _vp]),
    "elfrq_set_threads": (_i, [_vp, _i, _i64, C.c_uint64]),
    "elfrq_draw": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "elfsp_progress": (_i, [_vp, _vp]),
    "elfsp_set_pick_seed": (_i, [_vp, C.c_uint32]),
    "elfsp_game_actor": (_i, [_vp, _i]),
    "elfsp_begin_step": (_i, [_vp, _vp, _i64, C.POINTER(_i), _vp]),
    "elfsp_end_step": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "elfsp_last_rows": (_i, [_vp, C.POINTER(_i)]),
    "elfsp_set_request": (_i, [_vp, _i64, _i64, _f, _f, _i]),
    "elfsp_take_game_starts": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "elfsp_stats": (_i, [_vp, _vp]),
    "elfsp_games_finished": (_i64, [_vp]),
    "elfsp_search_log": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "elfsp_play": (_i, [_vp, _vp, _vp]),
    "elfsp_preload": (_i, [_vp, _vp, _i, _i, _vp]),
    "elfsp_restart": (_i, [_vp, _vp, _i, _vp]),
    "elfsp_finish": (_i, [_vp, _vp, _i, _i, _vp]),
    "elfsp_take_finished": (_i, [_vp, _vp, _i]),
    "elfsp_last_score": (_i, [_vp, _vp]),
    "elfsp_last_moves": (_i, [_vp, _vp]),
    "elfsp_records_pending": (_i, [_vp]),
    "elfsp_pop_record": (_i, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "elftrain_create": (_i, [_vp, _i, _i, _i, C.c_uint32, C.POINTER(_vp)]),
    "elftrain_destroy": (_i, [_vp]),
    "elftrain_capacity": (_i, [_vp]),
    "elftrain_max_moves": (_i, [_vp]),
    "elftrain_num_records": (_i, [_vp]),
    "elftrain_put": (_i, [_vp, _i, _vp, _i, _f, _i64, _vp, _i, _vp, _i]),
    "elftrain_put_async": (_i, [_vp, _i, _vp, _i, _f, _i64, _vp, _i, _vp, _i, _vp]),
    "elftrain_set_keep_states": (_i, [_vp, _i]),
    "elftrain_draw": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "elftrain_extract": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "elfrec_coords_to_sgfstr": (_i, [_i, _vp, _i, _vp, _sz]),
    "elfrec_sgfstr_to_coords": (_i, [_i, C.c_char_p, _vp, _i]),
    "elfrec_record_to_json": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _i, _i, C.c_uint64, C.c_uint64, _vp, _sz]),
    "elfrec_record_to_json2": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _f, _i, _i, C.c_uint64, C.c_uint64, _vp, _sz]),
    "elfrec_quantise_policy": (_i, [_i, _vp, _vp, _i, _vp]),
    "elfnet_bias_act_f16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "elfnet_bias_act_bf16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "elfgo_set_device": (_i, [_i]),
    "elfgo_get_device": (_i, [C.POINTER(_i)]),
    "elfgo_mem_info": (_i, [_i, C.POINTER(_sz), C.POINTER(_sz)]),
    "elfgo_pointer_kind": (_i, [_vp, C.POINTER(_i)]),
    "elfgo_memcpy2d_async": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "elfgo_stream_sync": (_i, [_vp]),
    "elfgo_malloc": (_i, [C.POINTER(_vp), _sz]),
    "elfgo_free": (_i, [_vp]),
    "elfgo_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "elfgo_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "elfgo_error_string": (C.c_char_p, [_i]),
    "elfgo_version": (C.c_char_p, []),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "elf_amd: %s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C elf_amd/csrc). There is no CPU fallback." % LIB_PATH
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the .so is stale: fail loudly
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class ElfGoError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise ElfGoError("libelf_amd status %d: %s" % (status, lib().elfgo_error_string(status).decode()))
