"""Host-side mirror (ctypes) of libconnect6_amd.so -- the MI355X-native Connect6 self-play engine.

The product is the HIP library next to this file (built by csrc/Makefile, see
__graft_entry__.build()); this module only binds its C ABI (include/connect6_amd.h).
There is NO CPU fallback: importing works without a GPU (so that symbols can be checked),
every compute call raises C6Error when the library or a gfx950 device is missing.

The directory name has a hyphen, so load it with `load_package()` from c6amd_loader.py or
`importlib` (tests/conftest.py, bench.py and __graft_entry__.py do).

Reference interfaces mirrored here:
  self_play / default_param   Connect6/pybind/pyconnect6/__init__.py:9-56
  Param builder               Connect6/cppbind/headers/connect6.hpp:272-309
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C6_LIB_PATH") or os.path.join(_HERE, "libconnect6_amd.so")   # override: A/B builds (tools/)

BLACK, NONE, WHITE = -1, 0, 1
EVAL_RANDOM, EVAL_UNIFORM, EVAL_NET, EVAL_CALLBACK, EVAL_STRESS = 1, 2, 3, 4, 5
OK, ERR_NO_GPU, ERR_BAD_ARG, ERR_OVERFLOW, ERR_HIP, ERR_INVALID_POSITION, ERR_ALREADY_SET = range(7)
# arithmetic of the 3x3 convolutions (include/connect6_amd.h C6_PREC_*).  DEFAULT = FP32: f32 like the reference net, evaluated as
# Winograd F(4x4,3x3) (256 channels) / F(2x2,3x3) (64, 128) on the FP32 matrix cores; FP32_DIRECT: the same arithmetic as a direct implicit GEMM
PREC_DEFAULT, PREC_FP32, PREC_BF16X3, PREC_BF16 = 0, 1, 3, 4      # PREC_BF16: relaxed tolerance (~1e-3), not parity grade
PREC_F16F8 = 5      # opt-in: fp16 main term + MX-fp8 cross terms, bounded range (guarded: C6_ERR_RANGE)
PREC_FP32_DIRECT = 6
ERR_OVERFLOW, ERR_RANGE = 3, 7
REF_N = 15

DROPIN_SYMBOLS = ["cpp_play", "cpp_self_play", "cpp_play_with", "test_new_raw_path", "test_with_raw_path",
                  "test_echo_raw_path", "test_with_raw_play_result", "test_echo_raw_play_result", "test_with_raw_vec",
                  "test_echo_raw_vec", "test_echo_cppeval", "test_cpp_policy"]
ENGINE_SYMBOLS = ["c6_last_error", "c6_version", "c6_rng_spec", "c6_device_count", "c6_default_config", "c6_engine_create",
                  "c6_engine_destroy", "c6_net_blob_floats", "c6_net_blob_header_floats", "c6_net_blob_write_header",
                  "c6_net_blob_check_header", "c6_engine_load_weights", "c6_engine_set_callback",
                  "c6_engine_step", "c6_engine_run", "c6_engine_maintain", "c6_engine_sync", "c6_engine_stats", "c6_engine_pool_audit", "c6_engine_fetch_games",
                  "c6_tuple_stride", "c6_engine_set_tuple_buffer", "c6_engine_swap_tuple_buffer", "c6_engine_stream", "c6_engine_enable_tuples", "c6_engine_fetch_tuples",
                  "c6_engine_root", "c6_engine_profile", "c6_engine_profile_read", "c6_board_step", "c6_search_winner",
                  "c6_legal_mask", "c6_augment8", "c6_recover8", "c6_net_forward", "c6_net_forward_prec", "c6_dirichlet_noise",
                  "c6_bench_conv", "c6_bench_board", "c6_bench_board_scan", "c6_pool_ring_selftest", "c6_conv_pack_m8_bytes", "c6_conv_pack_m8", "c6_replay_create", "c6_replay_create_on_device", "c6_replay_destroy", "c6_replay_push_dev", "c6_replay_push_host",
                  "c6_replay_clear", "c6_replay_size", "c6_replay_sample", "c6_engine_push_tuples", "c6_replay_sample_device",
                  "c6_replay_board_size", "c6_trainer_create", "c6_trainer_destroy", "c6_trainer_set_lr", "c6_trainer_load_weights",
                  "c6_trainer_export_weights", "c6_trainer_step", "c6_trainer_loss", "c6_trainer_step_replay", "c6_trainer_gradients",
                  "c6_trainer_last_error", "c6_trainer_set_precision", "c6_conv_wgrad_op", "c6_engine_boards", "c6_engine_play_match", "c6_engine_apply_moves", "c6_engine_last_moves",
                  "c6_rollout_create", "c6_rollout_destroy", "c6_rollout_reset", "c6_rollout_next", "c6_rollout_slot_info",
                  "c6_rollout_node", "c6_rollout_bench_playouts",
                  "c6_group_create", "c6_group_destroy", "c6_group_size", "c6_group_engine", "c6_group_member_info", "c6_group_uses_rccl",
                  "c6_group_load_weights", "c6_group_set_callback", "c6_group_step", "c6_group_run", "c6_group_sync", "c6_group_stats",
                  "c6_group_fetch_games", "c6_group_enable_tuples", "c6_group_pool_begin", "c6_group_pool_end", "c6_group_pool_tuples",
                  "c6_group_pooled", "c6_group_fetch_pooled", "c6_group_push_pooled"]


class C6Error(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("board_size", C.c_int), ("num_games", C.c_int), ("num_simulation", C.c_int), ("epsilon", C.c_float),
                ("dirichlet_alpha", C.c_double), ("c_puct", C.c_float), ("seed", C.c_uint64),
                ("game_uid_base", C.c_uint32), ("game_uid_stride", C.c_uint32), ("evaluator", C.c_int),
                ("net_channels", C.c_int), ("net_blocks", C.c_int), ("records_per_game", C.c_int), ("recycle", C.c_int),
                ("stagger", C.c_int), ("device", C.c_int), ("net_chunk_boards", C.c_int), ("net_precision", C.c_int),
                ("max_records_per_game", C.c_int), ("stress_every", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("rounds", C.c_uint64), ("simulations", C.c_uint64), ("select_steps", C.c_uint64), ("stones", C.c_uint64),
                ("games_finished", C.c_uint64), ("tuples", C.c_uint64), ("live_games", C.c_uint32),
                ("overflow_games", C.c_uint32), ("max_records_used", C.c_uint32), ("max_depth", C.c_uint32),
                ("abandoned_games", C.c_uint64), ("records_per_game", C.c_uint32), ("pool_grows", C.c_uint32),
                ("pool_pages", C.c_uint32), ("pool_pages_free", C.c_uint32), ("pool_pages_low", C.c_uint32), ("cap_raises", C.c_uint32),
                ("parked_games", C.c_uint32), ("max_pages_held", C.c_uint32), ("pool_bytes", C.c_uint64)]


class RootInfo(C.Structure):
    _fields_ = [("turn", C.c_int), ("remain", C.c_int), ("stones", C.c_int), ("status", C.c_int), ("winner", C.c_int),
                ("root_visit", C.c_int), ("root_value", C.c_float), ("root_q_sum", C.c_float), ("root_q_value", C.c_float),
                ("nsearch", C.c_uint32), ("records_used", C.c_int)]


class Profile(C.Structure):
    _fields_ = [("conv_ms", C.c_double), ("conv_launches", C.c_uint64), ("conv_flops", C.c_double), ("select_ms", C.c_double),
                ("expand_ms", C.c_double), ("eval_other_ms", C.c_double), ("move_ms", C.c_double),
                ("select_launches", C.c_uint64), ("expand_launches", C.c_uint64), ("conv_pre_ms", C.c_double),
                ("conv_exec_flops", C.c_double), ("conv_post_ms", C.c_double)]


class RolloutInfo(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("edges", C.c_uint32), ("searches", C.c_uint32), ("uid", C.c_uint32)]


class Path(C.Structure):  # cppbind/.../rawobj/mod.rs:29-36
    _fields_ = [("turn", C.c_int), ("board", (C.c_int * REF_N) * REF_N), ("row", C.c_int), ("col", C.c_int)]


class PlayResult(C.Structure):  # rawobj/mod.rs:38-43
    _fields_ = [("winner", C.c_int), ("paths", C.POINTER(Path)), ("len", C.c_int)]


class Vec(C.Structure):  # rawobj/mod.rs:45-50
    _fields_ = [("vec", C.POINTER(PlayResult)), ("len", C.c_int)]


class VecInt(C.Structure):
    _fields_ = [("vec", C.POINTER(C.c_int)), ("len", C.c_int)]


class VecFloat(C.Structure):
    _fields_ = [("vec", C.POINTER(C.c_float)), ("len", C.c_int)]


CALLBACK = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int)
POLICY_CALLBACK = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_int))
# ctypes callbacks cannot return POINTER(T): the allocators return the address as void*
ALLOC_PATH = C.CFUNCTYPE(C.c_void_p, C.c_int)
ALLOC_RESULT = C.CFUNCTYPE(C.c_void_p, C.c_int)
ALLOC_INT = C.CFUNCTYPE(C.c_void_p, C.c_int)
ALLOC_FLOAT = C.CFUNCTYPE(C.c_void_p, C.c_int)

_lib = None


def lib():
    """Load libconnect6_amd.so; fail loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise C6Error(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    i8p, u8p, f32p, i32p, u32p, i16p = (C.POINTER(t) for t in (C.c_int8, C.c_uint8, C.c_float, C.c_int32, C.c_uint32, C.c_int16))
    L.c6_last_error.restype = C.c_char_p
    L.c6_version.restype = C.c_char_p
    if hasattr(L, "c6_rng_spec"):      # (absent from the A/B libraries built from earlier rounds' sources: tools/ab_many.sh)
        L.c6_rng_spec.restype = C.c_char_p
    L.c6_default_config.argtypes = [C.POINTER(Config)]
    L.c6_engine_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.c6_engine_destroy.argtypes = [C.c_void_p]
    L.c6_net_blob_floats.restype = C.c_size_t
    L.c6_net_blob_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.c6_net_blob_header_floats.restype = C.c_size_t
    L.c6_net_blob_write_header.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int, C.c_int]
    L.c6_net_blob_check_header.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int, C.c_int]
    L.c6_engine_load_weights.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.c6_engine_set_callback.argtypes = [C.c_void_p, CALLBACK]
    L.c6_engine_step.argtypes = [C.c_void_p, C.c_int]
    L.c6_engine_run.argtypes = [C.c_void_p, C.c_long]
    L.c6_engine_sync.argtypes = [C.c_void_p]
    L.c6_engine_maintain.argtypes = [C.c_void_p]
    L.c6_engine_apply_moves.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.c6_engine_last_moves.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.c6_engine_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    if hasattr(L, "c6_engine_pool_audit"):
        L.c6_engine_pool_audit.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 12)]
    L.c6_engine_play_match.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_long]
    L.c6_engine_boards.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int8)]
    L.c6_rollout_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.c6_rollout_destroy.argtypes = [C.c_void_p]
    L.c6_rollout_reset.argtypes = [C.c_void_p, u8p]
    L.c6_rollout_next.argtypes = [C.c_void_p, i8p, u8p, i32p, C.c_int]
    L.c6_rollout_slot_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(RolloutInfo)]
    L.c6_rollout_node.argtypes = [C.c_void_p, C.c_int, i8p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.c_int, i32p, i32p, i32p]
    L.c6_rollout_bench_playouts.argtypes = [C.c_void_p, i8p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), i8p]
    L.c6_engine_fetch_games.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), i32p, i32p, u32p, i16p]
    L.c6_tuple_stride.restype = C.c_size_t
    L.c6_tuple_stride.argtypes = [C.c_int]
    L.c6_engine_set_tuple_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.c6_engine_enable_tuples.argtypes = [C.c_void_p, C.c_size_t]
    L.c6_engine_swap_tuple_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.c6_engine_stream.argtypes = [C.c_void_p]
    L.c6_engine_stream.restype = C.c_void_p
    L.c6_engine_fetch_tuples.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.c6_engine_root.argtypes = [C.c_void_p, C.c_int, C.POINTER(RootInfo), i8p, i32p, f32p, f32p]
    L.c6_engine_profile.argtypes = [C.c_void_p, C.c_int]
    L.c6_engine_profile_read.argtypes = [C.c_void_p, C.POINTER(Profile), C.c_int]
    L.c6_board_step.argtypes = [C.c_int, C.c_int, i8p, i8p, i8p, i32p, i8p, i8p]
    L.c6_search_winner.argtypes = [C.c_int, C.c_int, i8p, i8p]
    L.c6_legal_mask.argtypes = [C.c_int, C.c_int, i8p, u8p]
    L.c6_augment8.argtypes = [C.c_int, C.c_int, i8p, i8p]
    L.c6_recover8.argtypes = [C.c_int, C.c_int, i8p, f32p, f32p, f32p, f32p]
    L.c6_net_forward.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.c_size_t, C.c_int, i8p, i8p, f32p, f32p]
    L.c6_dirichlet_noise.argtypes = [C.c_int, C.c_int, C.c_double, u32p, u32p, u32p, u32p, i8p, f32p]
    L.c6_net_forward_prec.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_size_t, C.c_int, i8p, i8p, f32p, f32p]
    L.c6_bench_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
    L.c6_bench_board.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.c6_bench_board_scan.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.c6_conv_pack_m8_bytes.argtypes = [C.c_int]
    L.c6_conv_pack_m8_bytes.restype = C.c_size_t
    L.c6_conv_pack_m8.argtypes = [C.c_int, f32p, C.POINTER(C.c_uint8), C.c_size_t]
    L.c6_replay_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.c6_replay_destroy.argtypes = [C.c_void_p]
    L.c6_replay_push_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.c6_replay_push_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.c6_replay_clear.argtypes = [C.c_void_p]
    L.c6_engine_push_tuples.argtypes = [C.c_void_p, C.c_void_p]
    L.c6_replay_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.c6_replay_sample.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, f32p, f32p, f32p, f32p, f32p]
    L.c6_trainer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_void_p)]
    L.c6_trainer_destroy.argtypes = [C.c_void_p]
    L.c6_trainer_set_lr.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.c6_trainer_set_precision.argtypes = [C.c_void_p, C.c_int]
    L.c6_conv_wgrad_op.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    L.c6_trainer_load_weights.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.c6_trainer_export_weights.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.c6_trainer_gradients.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.c6_trainer_step.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_int, f32p]
    L.c6_trainer_loss.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_int, f32p]
    L.c6_trainer_step_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, f32p]
    L.c6_trainer_last_error.restype = C.c_char_p
    if hasattr(L, "c6_group_create"):      # (absent from A/B libraries built from earlier rounds' sources)
        L.c6_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.c6_group_destroy.argtypes = [C.c_void_p]
        L.c6_group_size.argtypes = [C.c_void_p]
        L.c6_group_engine.argtypes = [C.c_void_p, C.c_int]
        L.c6_group_engine.restype = C.c_void_p
        L.c6_group_member_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), u32p, C.POINTER(C.c_int)]
        L.c6_group_uses_rccl.argtypes = [C.c_void_p]
        L.c6_group_load_weights.argtypes = [C.c_void_p, f32p, C.c_size_t]
        L.c6_group_set_callback.argtypes = [C.c_void_p, CALLBACK, C.c_int]
        L.c6_group_step.argtypes = [C.c_void_p, C.c_int]
        L.c6_group_run.argtypes = [C.c_void_p, C.c_long]
        L.c6_group_sync.argtypes = [C.c_void_p]
        L.c6_group_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.c6_group_fetch_games.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), i32p, i32p, u32p, i16p]
        L.c6_group_enable_tuples.argtypes = [C.c_void_p, C.c_size_t]
        L.c6_group_pool_begin.argtypes = [C.c_void_p]
        L.c6_group_pool_end.argtypes = [C.c_void_p, u32p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
        L.c6_group_pool_tuples.argtypes = [C.c_void_p, u32p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
        L.c6_group_pooled.argtypes = [C.c_void_p, C.c_int]
        L.c6_group_pooled.restype = C.c_void_p
        L.c6_group_fetch_pooled.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
        L.c6_group_push_pooled.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.cpp_self_play.restype = Vec
    L.cpp_self_play.argtypes = [CALLBACK, ALLOC_PATH, ALLOC_RESULT, C.c_int, C.c_float, C.c_double, C.c_float, C.c_bool, C.c_int]
    L.cpp_play.restype = Vec
    L.cpp_play.argtypes = [POLICY_CALLBACK, ALLOC_PATH, ALLOC_RESULT, C.c_bool, C.c_int]
    L.test_new_raw_path.restype = Path
    L.test_with_raw_path.restype = Path
    L.test_echo_raw_path.restype = Path
    L.test_echo_raw_path.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.test_with_raw_play_result.restype = PlayResult
    L.test_with_raw_play_result.argtypes = [ALLOC_PATH]
    L.test_echo_raw_play_result.restype = PlayResult
    L.test_echo_raw_play_result.argtypes = [C.c_int, C.POINTER(Path), C.c_int, ALLOC_PATH]
    L.test_with_raw_vec.restype = VecInt
    L.test_with_raw_vec.argtypes = [ALLOC_INT]
    L.test_echo_raw_vec.restype = VecInt
    L.test_echo_raw_vec.argtypes = [C.POINTER(C.c_int), C.c_int, ALLOC_INT]
    L.test_echo_cppeval.restype = VecFloat
    L.test_echo_cppeval.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, CALLBACK, ALLOC_FLOAT]
    L.test_cpp_policy.restype = VecInt
    L.test_cpp_policy.argtypes = [C.POINTER(C.c_float), POLICY_CALLBACK, ALLOC_INT]
    _lib = L
    return L


def _chk(rc):
    if rc != OK:
        msg = lib().c6_last_error().decode() or lib().c6_trainer_last_error().decode()
        raise C6Error(f"libconnect6_amd error {rc}: {msg}")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pool_ring_selftest(ids=1000, waves=4096, tries=4, launches=50, seed=1):
    """c6_pool_ring_selftest -> dict(dup, over, min_popped)"""
    d, ov, mp = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _chk(lib().c6_pool_ring_selftest(ids, waves, tries, launches, seed, C.byref(d), C.byref(ov), C.byref(mp)))
    return dict(dup=d.value, over=ov.value, min_popped=mp.value)


def rng_spec():
    """c6_rng_spec: the random-stream specification the library implements (libraries from before round 4 do not say: they are C6RNG v4)"""
    L = lib()
    return L.c6_rng_spec().decode() if hasattr(L, "c6_rng_spec") else "C6RNG v4"


def device_count():
    return lib().c6_device_count()


# ---------------------------------------------------------------------------------------------
# caller-side allocators (connect6.hpp:17-23 `new T[n]`): memory is owned by the Python side
# ---------------------------------------------------------------------------------------------
class _Arena:
    def __init__(self):
        self.keep = []
        self.alloc_path = ALLOC_PATH(lambda n: self._mk(Path, n))
        self.alloc_result = ALLOC_RESULT(lambda n: self._mk(PlayResult, n))
        self.alloc_int = ALLOC_INT(lambda n: self._mk(C.c_int, n))
        self.alloc_float = ALLOC_FLOAT(lambda n: self._mk(C.c_float, n))

    def _mk(self, t, n):
        arr = (t * max(int(n), 1))()
        self.keep.append(arr)
        return C.addressof(arr)


def _vec_to_py(v):
    """GameResult::from_vec (connect6.hpp:251-263) -> [(winner, [(turn, board[15][15], (row, col)), ...])] with numpy boards"""
    out = []
    for i in range(v.len):
        r = v.vec[i]
        paths = []
        for k in range(r.len):
            p = r.paths[k]
            board = np.ctypeslib.as_array(p.board).astype(np.int8).reshape(REF_N, REF_N).copy()
            paths.append((p.turn, board, (p.row, p.col)))
        out.append((r.winner, paths))
    return out


def _result_to_ref(r):
    """RunResultWrapper / PathWrapper (pybind/pybind_impl/mod.rs:66-105): `(winner: int, [(turn: int, board: list of N*N ints
    in row-major order (pylist_from_board, :29-37), (row, col)), ...])` -- plain Python objects, as rust-cpython builds them"""
    path = []
    for k in range(r.len):
        p = r.paths[k]
        path.append((int(p.turn), np.ctypeslib.as_array(p.board).reshape(-1).tolist(), (int(p.row), int(p.col))))
    return int(r.winner), path


def board_size():
    """pyconnect6.board_size (pybind/pyconnect6/__init__.py:5-6): the drop-in boundary is the reference's 15x15 build"""
    return REF_N


def default_param():
    """pyconnect6.default_param (Connect6/pybind/pyconnect6/__init__.py:46-55)"""
    return {"num_simulation": 800, "epsilon": 0.25, "dirichlet_alpha": 0.03, "c_puct": 1, "debug": False, "num_game_thread": 1}


def param_to_tuple(param):
    """pyconnect6.param_to_tuple (pybind/pyconnect6/__init__.py:58-64); a missing key is a KeyError, as there"""
    return (param["num_simulation"], param["epsilon"], param["dirichlet_alpha"], param["c_puct"], param["debug"],
            param["num_game_thread"])


def dump_param(path, param=None):
    """pyconnect6.dump_param (pybind/pyconnect6/__init__.py:67-73): MCTS hyper-parameters as `<path>_mcts.json`"""
    import json
    if param is None:
        param = default_param()
    with open(path + "_mcts.json", "w") as f:
        f.write(json.dumps(param))


def load_param(path):
    """pyconnect6.load_param (pybind/pyconnect6/__init__.py:76-78)"""
    import json
    with open(path + "_mcts.json") as f:
        return json.loads(f.read())


def _cpp_self_play(tramp, param):
    cb = CALLBACK(tramp)
    arena = _Arena()
    ns, eps, alpha, cp, debug, threads = param_to_tuple(param)
    v = lib().cpp_self_play(cb, arena.alloc_path, arena.alloc_result, int(ns), float(eps), float(alpha), float(cp), bool(debug), int(threads))
    return v, arena


def self_play(policy, param=None):
    """pyconnect6.self_play (pybind/pyconnect6/__init__.py:9-35 over libpyconnect6/src/lib.rs:104-153) through the reference C
    ABI cpp_self_play, with the reference's Python objects on both sides:

    `policy(turn: int, boards: list[len] of list[N*N] of int in {-1, 0, 1}) -> (values[len], probs[len][N*N])` -- PyEval
    (pybind/pyeval/mod.rs: `pylist_from_multiple` in, anything iterable back; numpy arrays are fine, as in
    Connect6/sample/python/random_policy.py:22-26);

    returns `(winner, path)` when `param['num_game_thread'] == 1` (lib.rs:119-128) and a tuple of `(winner, path)` -- one per
    game, in any order -- otherwise (lib.rs:129-152); `path = [(turn, board: list[N*N] of int, (row, col)), ...]`
    (pybind_impl/mod.rs:66-105).  `self_play_np` is the same call with numpy arrays and always a list of games."""
    if param is None:
        param = default_param()
    cells = REF_N * REF_N

    def tramp(player, values, policies, n):
        pol = np.ctypeslib.as_array(policies, shape=(n, cells))
        val = np.ctypeslib.as_array(values, shape=(n,))
        v, p = policy(int(player), pol.astype(np.int32).tolist())
        val[:] = np.asarray(v, dtype=np.float32).reshape(n)
        pol[:] = np.asarray(p, dtype=np.float32).reshape(n, cells)

    v, arena = _cpp_self_play(tramp, param)
    games = tuple(_result_to_ref(v.vec[i]) for i in range(v.len))
    if param["num_game_thread"] == 1:
        return games[0]
    return games


def self_play_np(policy, param=None):
    """the numpy face of the same boundary (not in the reference): `policy(turn, boards: int8 ndarray [len, 225]) -> (values,
    probs)`; returns a LIST of `(winner, [(turn, board int8[15, 15], (row, col)), ...])`, one per game, whatever
    `num_game_thread` is; missing keys of `param` take their defaults"""
    prm = default_param()
    prm.update(param or {})

    def tramp(player, values, policies, n):
        pol = np.ctypeslib.as_array(policies, shape=(n, REF_N * REF_N))
        val = np.ctypeslib.as_array(values, shape=(n,))
        v, p = policy(player, pol.astype(np.int8))
        val[:] = np.asarray(v, dtype=np.float32).reshape(n)
        pol[:] = np.asarray(p, dtype=np.float32).reshape(n, REF_N * REF_N)

    v, arena = _cpp_self_play(tramp, prm)
    return _vec_to_py(v)


class RandomPolicy:
    """policy/random_policy/mod.rs:39-47 as a `play()` policy: a uniformly random empty cell, None on a full board
    (`possible.choose(&mut thread_rng())`; the reference's generator is unseeded, this one takes a seed)"""

    def __init__(self, seed=None):
        self.rng = np.random.default_rng(seed)

    def __call__(self, board):
        empty = np.flatnonzero(np.asarray(board).reshape(-1) == 0)
        if empty.size == 0:
            return None
        c = int(self.rng.choice(empty))
        n = int(round(np.sqrt(np.asarray(board).size)))
        return c // n, c % n


def play(policy, debug=False, num_game_thread=1):
    """Connect6::play (connect6.hpp:311-323) through cpp_play: policy(board[15,15] float) -> (row, col) or None"""

    def tramp(board, position):
        b = np.ctypeslib.as_array(board, shape=(REF_N, REF_N))
        pos = policy(b)
        if pos is not None:
            position[0], position[1] = int(pos[0]), int(pos[1])

    cb = POLICY_CALLBACK(tramp)
    arena = _Arena()
    v = lib().cpp_play(cb, arena.alloc_path, arena.alloc_result, bool(debug), num_game_thread)
    return _vec_to_py(v)


def play_with(policy, param=None):
    """pyconnect6.play_with (pybind/pyconnect6/__init__.py:38-43 over lib.rs:170-192) through cpp_play_with: AlphaZero over
    `policy` (the evaluator callable of self_play) plays Black, a human on stdin / stdout plays White ("aB" = row a, column B).
    Returns `(winner, path)` in self_play's shapes."""
    if param is None:
        param = default_param()
    cells = REF_N * REF_N

    def tramp(player, values, policies, n):
        pol = np.ctypeslib.as_array(policies, shape=(n, cells))
        val = np.ctypeslib.as_array(values, shape=(n,))
        v, p = policy(int(player), pol.astype(np.int32).tolist())
        val[:] = np.asarray(v, dtype=np.float32).reshape(n)
        pol[:] = np.asarray(p, dtype=np.float32).reshape(n, cells)

    cb = CALLBACK(tramp)
    arena = _Arena()
    L = lib()
    L.cpp_play_with.restype = PlayResult
    L.cpp_play_with.argtypes = [CALLBACK, ALLOC_PATH, C.c_int, C.c_float, C.c_double, C.c_float]
    ns, eps, alpha, cp = param_to_tuple(param)[:-2]
    r = L.cpp_play_with(cb, arena.alloc_path, int(ns), float(eps), float(alpha), float(cp))
    return _result_to_ref(r)


# ---------------------------------------------------------------------------------------------
# engine
# ---------------------------------------------------------------------------------------------
class Engine:
    """G concurrent AlphaZero self-play games resident on one MI355X (c6_engine)."""

    def __init__(self, board_size=15, num_games=1, num_simulation=800, epsilon=0.25, dirichlet_alpha=0.03, c_puct=1.0,
                 seed=0, evaluator=EVAL_RANDOM, net_channels=256, net_blocks=10, records_per_game=0, recycle=False,
                 stagger=False, device=0, game_uid_base=0, game_uid_stride=0, net_chunk_boards=0, net_precision=PREC_DEFAULT,
                 max_records_per_game=0, stress_every=0):
        L = lib()
        cfg = Config()
        L.c6_default_config(C.byref(cfg))
        cfg.board_size, cfg.num_games, cfg.num_simulation = board_size, num_games, num_simulation
        cfg.epsilon, cfg.dirichlet_alpha, cfg.c_puct, cfg.seed = epsilon, dirichlet_alpha, c_puct, seed
        cfg.evaluator, cfg.net_channels, cfg.net_blocks = evaluator, net_channels, net_blocks
        cfg.records_per_game, cfg.recycle, cfg.stagger, cfg.device = records_per_game, int(recycle), int(stagger), device
        cfg.game_uid_base, cfg.game_uid_stride, cfg.net_chunk_boards = game_uid_base, game_uid_stride, net_chunk_boards
        cfg.net_precision = net_precision
        cfg.max_records_per_game = max_records_per_game
        cfg.stress_every = stress_every
        self.cfg = cfg
        self.n, self.cells, self.G = board_size, board_size * board_size, num_games
        self.h = C.c_void_p()
        self._cb = None
        _chk(L.c6_engine_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().c6_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _chk(lib().c6_engine_load_weights(self.h, _p(blob, C.c_float), blob.size))

    def set_callback(self, fn):
        """fn(player, boards int8 [len, N*N]) -> (values [len], policies [len, N*N])"""
        cells = self.cells

        def tramp(player, values, policies, n):
            pol = np.ctypeslib.as_array(policies, shape=(n, cells))
            val = np.ctypeslib.as_array(values, shape=(n,))
            v, p = fn(player, pol.astype(np.int8))
            val[:] = np.asarray(v, dtype=np.float32).reshape(n)
            pol[:] = np.asarray(p, dtype=np.float32).reshape(n, cells)

        self._cb = CALLBACK(tramp)
        _chk(lib().c6_engine_set_callback(self.h, self._cb))

    def step(self, rounds=1):
        _chk(lib().c6_engine_step(self.h, rounds))

    def run(self, max_rounds=0):
        _chk(lib().c6_engine_run(self.h, max_rounds))

    def sync(self):
        _chk(lib().c6_engine_sync(self.h))

    def maintain(self):
        """synchronise; raise the per-game caps / map more memory into the shared record pool if games are waiting for it or it is
        running low (c6_engine_run does this by itself; callers of step() call it now and then)"""
        _chk(lib().c6_engine_maintain(self.h))

    def last_moves(self):
        """cells chosen in the last move phase: cell, -1 = no move (draw), -2 = the game did not move"""
        p = np.zeros(self.G, np.int32)
        _chk(lib().c6_engine_last_moves(self.h, _p(p, C.c_int32)))
        return p

    def apply_moves(self, pos):
        """stones from outside the engine's policy (MultiPolicy / the human of play_with): pos[g] = cell or -2 for none"""
        p = np.ascontiguousarray(pos, dtype=np.int32).reshape(-1)
        if p.size != self.G:
            raise C6Error("apply_moves needs one entry per game")
        _chk(lib().c6_engine_apply_moves(self.h, _p(p, C.c_int32)))

    def play_match(self, engine_colour, opponent="random", rollout=None, opp_seed=0, max_stones=0):
        """evaluation games of this engine's search (as `engine_colour`: -1 Black, +1 White) against RandomPolicy ("random") or a
        DefaultPolicy object ("rollout", rollout=DefaultPolicy(...)) on the device; results: fetch_games()"""
        kind = {"random": 1, "rollout": 2}[opponent]
        _chk(lib().c6_engine_play_match(self.h, engine_colour, kind, rollout.h if rollout is not None else None, opp_seed, max_stones))

    def boards(self):
        """K1's canonical state: (boards int8 [G, N*N], turn [G], remain [G])"""
        b = np.zeros((self.G, self.cells), np.int8)
        t = np.zeros(self.G, np.int8)
        r = np.zeros(self.G, np.int8)
        _chk(lib().c6_engine_boards(self.h, _p(b, C.c_int8), _p(t, C.c_int8), _p(r, C.c_int8)))
        return b, t, r

    def stats(self):
        s = Stats()
        _chk(lib().c6_engine_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def pool_audit(self):
        """c6_engine_pool_audit: {pages,chunks}_{total,held,free,dup,leaked,bad}; dup / leaked / bad are 0 on a sound engine"""
        a = (C.c_uint32 * 12)()
        _chk(lib().c6_engine_pool_audit(self.h, C.byref(a)))
        names = [f"{k}_{f}" for k in ("pages", "chunks") for f in ("total", "held", "free", "dup", "leaked", "bad")]
        return dict(zip(names, [int(x) for x in a]))

    def fetch_games(self, cap=None):
        cap = cap or 4 * self.G
        winners = np.zeros(cap, np.int32)
        lens = np.zeros(cap, np.int32)
        uids = np.zeros(cap, np.uint32)
        moves = np.zeros((cap, self.cells), np.int16)
        cnt = C.c_int()
        _chk(lib().c6_engine_fetch_games(self.h, cap, C.byref(cnt), _p(winners, C.c_int32), _p(lens, C.c_int32),
                                         _p(uids, C.c_uint32), _p(moves, C.c_int16)))
        k = cnt.value
        return [dict(winner=int(winners[i]), uid=int(uids[i]), moves=moves[i, :lens[i]].copy()) for i in range(k)]

    def enable_tuples(self, capacity):
        _chk(lib().c6_engine_enable_tuples(self.h, capacity))

    def set_tuple_buffer(self, dev_ptr, capacity, count_dev_ptr):
        _chk(lib().c6_engine_set_tuple_buffer(self.h, C.c_void_p(dev_ptr), capacity, C.c_void_p(count_dev_ptr)))

    def swap_tuple_buffer(self, dev_ptr, capacity, count_dev_ptr):
        """set_tuple_buffer without draining the engine's stream (double-buffered exchange, shard.TuplePool)"""
        _chk(lib().c6_engine_swap_tuple_buffer(self.h, C.c_void_p(dev_ptr), capacity, C.c_void_p(count_dev_ptr)))

    def stream_ptr(self):
        """the engine's HIP stream as an integer (torch.cuda.ExternalStream(ptr))"""
        return int(lib().c6_engine_stream(self.h) or 0)

    def push_tuples(self, replay):
        """device-to-device hand-off of the emitted tuples to a ReplayBuffer"""
        _chk(lib().c6_engine_push_tuples(self.h, replay.h))

    def fetch_tuples(self, cap):
        stride = lib().c6_tuple_stride(self.n)
        buf = np.zeros((cap, stride), np.uint8)
        cnt = C.c_size_t()
        _chk(lib().c6_engine_fetch_tuples(self.h, cap, C.byref(cnt), buf.ctypes.data_as(C.c_void_p)))
        return decode_tuples(buf[:cnt.value], self.n)

    def root(self, g=0):
        info = RootInfo()
        board = np.zeros(self.cells, np.int8)
        visits = np.zeros(self.cells, np.int32)
        q = np.zeros(self.cells, np.float32)
        prior = np.zeros(self.cells, np.float32)
        _chk(lib().c6_engine_root(self.h, g, C.byref(info), _p(board, C.c_int8), _p(visits, C.c_int32), _p(q, C.c_float),
                                  _p(prior, C.c_float)))
        d = {k: getattr(info, k) for k, _ in RootInfo._fields_}
        d.update(board=board.reshape(self.n, self.n), visits=visits, q=q, prior=prior)
        return d

    def profile(self, enable=True):
        _chk(lib().c6_engine_profile(self.h, int(enable)))

    def profile_read(self, reset=True):
        p = Profile()
        _chk(lib().c6_engine_profile_read(self.h, C.byref(p), int(reset)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}


class Group:
    """c6_group (include/connect6_amd.h part 6): one engine + one host thread per device behind the C ABI, games sharded by uid, training tuples
    pooled over the members by ncclAllGather (host-staged when members share a device) -- the library-side counterpart of AsyncAgent::run's
    thread pool + mpsc collection (agent/async_agent/mod.rs:100-137).  `devices`: HIP ordinals, repeats allowed; None = every visible device."""

    def __init__(self, devices=None, board_size=15, num_games=1, num_simulation=800, epsilon=0.25, dirichlet_alpha=0.03, c_puct=1.0, seed=0,
                 evaluator=EVAL_RANDOM, net_channels=256, net_blocks=4, records_per_game=0, recycle=False, stagger=False, game_uid_base=0,
                 net_chunk_boards=0, net_precision=0, max_records_per_game=0, stress_every=0):
        cfg = Config()
        lib().c6_default_config(C.byref(cfg))
        for k, v in dict(board_size=board_size, num_games=num_games, num_simulation=num_simulation, epsilon=epsilon, dirichlet_alpha=dirichlet_alpha,
                         c_puct=c_puct, seed=seed, evaluator=evaluator, net_channels=net_channels, net_blocks=net_blocks,
                         records_per_game=records_per_game, recycle=int(recycle), stagger=int(stagger), game_uid_base=game_uid_base,
                         net_chunk_boards=net_chunk_boards, net_precision=net_precision, max_records_per_game=max_records_per_game,
                         stress_every=stress_every).items():
            setattr(cfg, k, v)
        self.n, self.cells, self.G = board_size, board_size * board_size, num_games
        self.h = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices) if devices else None
        _chk(lib().c6_group_create(C.byref(cfg), dv, len(devices) if devices else 0, C.byref(self.h)))
        self.size = lib().c6_group_size(self.h)
        self.stride = int(lib().c6_tuple_stride(board_size))
        self._cb = None

    def close(self):
        if self.h:
            lib().c6_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def uses_rccl(self):
        return bool(lib().c6_group_uses_rccl(self.h))

    def member_info(self, r):
        dev, first, games = C.c_int(), C.c_uint32(), C.c_int()
        _chk(lib().c6_group_member_info(self.h, r, C.byref(dev), C.byref(first), C.byref(games)))
        return dict(device=dev.value, first_uid=first.value, games=games.value)

    def engine_root(self, member, g=0):
        """root of slot g of member `member` (c6_engine_root on the member's engine)"""
        e = Engine.__new__(Engine)
        e.h, e.n, e.cells = C.c_void_p(lib().c6_group_engine(self.h, member)), self.n, self.cells
        try:
            return Engine.root(e, g)
        finally:
            e.h = None      # borrowed: the group owns the engine

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        _chk(lib().c6_group_load_weights(self.h, _p(blob, C.c_float), blob.size))

    def set_callback(self, fn, concurrent=False):
        """fn(player, boards int8 [len, N*N]) -> (values [len], policies [len, N*N]) as Engine.set_callback; called from the members' host threads"""
        n2 = self.cells

        def tramp(player, values, policies, n):
            b = np.ctypeslib.as_array(policies, shape=(n, n2))
            v, p = fn(player, b.astype(np.int8))
            np.ctypeslib.as_array(values, shape=(n,))[:] = np.asarray(v, np.float32).reshape(n)
            b[:] = np.asarray(p, np.float32).reshape(n, n2)
        self._cb = CALLBACK(tramp)
        _chk(lib().c6_group_set_callback(self.h, self._cb, int(concurrent)))

    def step(self, rounds=1):
        _chk(lib().c6_group_step(self.h, rounds))

    def run(self, max_rounds=0):
        _chk(lib().c6_group_run(self.h, max_rounds))

    def sync(self):
        _chk(lib().c6_group_sync(self.h))

    def stats(self):
        s = Stats()
        _chk(lib().c6_group_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def fetch_games(self, cap=None):
        cap = cap or max(4 * self.G, 1024)
        winners, lens = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        uids, moves = np.zeros(cap, np.uint32), np.zeros((cap, self.cells), np.int16)
        cnt = C.c_int()
        _chk(lib().c6_group_fetch_games(self.h, cap, C.byref(cnt), _p(winners, C.c_int32), _p(lens, C.c_int32), _p(uids, C.c_uint32), _p(moves, C.c_int16)))
        return [dict(winner=int(winners[i]), uid=int(uids[i]), moves=moves[i, :lens[i]].copy()) for i in range(cnt.value)]

    def enable_tuples(self, capacity):
        _chk(lib().c6_group_enable_tuples(self.h, capacity))

    def pool_begin(self):
        _chk(lib().c6_group_pool_begin(self.h))

    def pool_end(self):
        counts = np.zeros(self.size, np.uint32)
        padded, dropped = C.c_size_t(), C.c_uint64()
        _chk(lib().c6_group_pool_end(self.h, _p(counts, C.c_uint32), C.byref(padded), C.byref(dropped)))
        return dict(counts=counts, padded=padded.value, dropped=dropped.value)

    def pool_tuples(self):
        self.pool_begin()
        return self.pool_end()

    def fetch_pooled(self, member=0, cap=None):
        """the valid records of member `member`'s copy of the last pool: uint8 [count, stride]"""
        cap = cap or self.G * self.cells
        buf = np.zeros((cap, self.stride), np.uint8)
        cnt = C.c_size_t()
        _chk(lib().c6_group_fetch_pooled(self.h, member, cap, C.byref(cnt), buf.ctypes.data_as(C.c_void_p)))
        return buf[:cnt.value]

    def push_pooled(self, member, replay):
        _chk(lib().c6_group_push_pooled(self.h, member, replay.h))


def decode_tuples(buf, n):
    """[state int8 N*N | pad | pi u16 N*N | ... | turn int8 | z int8] records -> dict of arrays"""
    cells = n * n
    buf = np.ascontiguousarray(buf)
    k = buf.shape[0]
    off = cells + (cells & 1)
    state = buf[:, :cells].view(np.int8).reshape(k, n, n)
    pi = buf[:, off:off + 2 * cells].copy().view(np.uint16).reshape(k, cells)
    turn = buf[:, -2].view(np.int8)
    z = buf[:, -1].view(np.int8)
    pos = buf[:, -4:-2].copy().view(np.uint16).reshape(k)
    return dict(state=state.copy(), pi=pi, turn=turn.copy(), z=z.copy(), pos=pos)


def encode_tuples(n, games):
    """[(winner, [(turn, board[n,n], (row, col)), ...], pi[len, n*n] or None), ...] -> records in the engine's tuple layout
    (what the engine emits for a finished game; used to feed a ReplayBuffer from host-side game records)"""
    cells = n * n
    off = cells + (cells & 1)
    stride = lib().c6_tuple_stride(n)
    recs = []
    for winner, path, pi in games:
        for k, (turn, board, (row, col)) in enumerate(path):
            r = np.zeros(stride, np.uint8)
            r[:cells] = np.asarray(board, dtype=np.int8).reshape(-1).view(np.uint8)
            if pi is not None:
                r[off:off + 2 * cells] = np.asarray(pi[k]).astype(np.uint16).view(np.uint8)
            r[-4:-2] = np.array([row * n + col], np.uint16).view(np.uint8)
            r[-2] = np.int8(turn).view(np.uint8)
            r[-1] = np.int8(winner).view(np.uint8)
            recs.append(r)
    return np.stack(recs) if recs else np.zeros((0, stride), np.uint8)


class ReplayBuffer:
    """device-resident ReplayBuffer (AlphaZero/cpp_weighted/ReplayBuffer.hpp:10-73) over engine tuple records"""

    def __init__(self, board_size, max_size):
        self.n, self.cells, self.max_size = board_size, board_size * board_size, max_size
        self.h = C.c_void_p()
        _chk(lib().c6_replay_create(board_size, max_size, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().c6_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def push_records(self, records):
        rec = np.ascontiguousarray(records, dtype=np.uint8)
        _chk(lib().c6_replay_push_host(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0]))

    def push_game(self, result, pi=None):
        """pyconnect6-shaped game result (winner, [(turn, board, (row, col)), ...]), py_weighted/buffer.py:22-32"""
        winner, path = result
        self.push_records(encode_tuples(self.n, [(winner, path, pi)]))

    def push_device(self, tuples_dev_ptr, count_dev_ptr, cap):
        _chk(lib().c6_replay_push_dev(self.h, C.c_void_p(tuples_dev_ptr), C.c_void_p(count_dev_ptr), cap))

    def clear(self):
        _chk(lib().c6_replay_clear(self.h))

    def size(self):
        nd, tail = C.c_size_t(), C.c_size_t()
        _chk(lib().c6_replay_size(self.h, C.byref(nd), C.byref(tail)))
        return nd.value, tail.value

    def __len__(self):
        return self.size()[0]

    def sample(self, num_sample, seed=0, draw_id=0, want_pi=False):
        w = np.zeros(num_sample, np.float32); p = np.zeros(num_sample, np.float32)
        b = np.zeros((num_sample, self.cells), np.float32); q = np.zeros(num_sample, np.float32)
        pi = np.zeros((num_sample, self.cells), np.float32) if want_pi else None
        _chk(lib().c6_replay_sample(self.h, num_sample, seed, draw_id, _p(w, C.c_float), _p(p, C.c_float), _p(b, C.c_float),
                                    _p(q, C.c_float), _p(pi, C.c_float) if want_pi else None))
        return (w, p, b, q, pi) if want_pi else (w, p, b, q)


class Trainer:
    """The train-loop body of AlphaZero/cpp_weighted/main.cpp:148-160 on the GPU: `step` = zero_grad + model->loss +
    backward + SGD step (WeightedPolicy in train mode, model.hpp:52-108).  Weights are the engine's blob."""

    def __init__(self, board_size, channels, blocks, batch, lr=1e-3, momentum=0.9, device=0, precision=PREC_DEFAULT):
        """precision: PREC_DEFAULT = PREC_FP32 (f32 like the reference trainer; Winograd-domain kernels at 256 channels), PREC_FP32_DIRECT
        (the same arithmetic on the direct kernels), PREC_BF16X3 (split-bf16 products: the fast option, gradients within ~1e-3)"""
        self.n, self.cells, self.C, self.blocks, self.batch = board_size, board_size * board_size, channels, blocks, batch
        self.nfloats = lib().c6_net_blob_floats(board_size, channels, blocks)
        self.h = C.c_void_p()
        _chk(lib().c6_trainer_create(board_size, channels, blocks, batch, lr, momentum, device, C.byref(self.h)))
        _chk(lib().c6_trainer_set_precision(self.h, precision))

    def close(self):
        if getattr(self, "h", None):
            lib().c6_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def load_weights(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.float32)
        _chk(lib().c6_trainer_load_weights(self.h, _p(b, C.c_float), b.size))

    def export_weights(self):
        b = np.zeros(self.nfloats, np.float32)
        _chk(lib().c6_trainer_export_weights(self.h, _p(b, C.c_float), b.size))
        return b

    def gradients(self):
        b = np.zeros(self.nfloats, np.float32)
        _chk(lib().c6_trainer_gradients(self.h, _p(b, C.c_float), b.size))
        return b

    def _batch(self, winners, players, boards, poses):
        w = np.ascontiguousarray(winners, dtype=np.float32).reshape(-1)
        p = np.ascontiguousarray(players, dtype=np.float32).reshape(-1)
        b = np.ascontiguousarray(boards, dtype=np.float32).reshape(-1)
        q = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1)
        if not (w.size == p.size == q.size == self.batch and b.size == self.batch * self.cells):
            raise C6Error("batch shapes do not match the trainer's mini-batch size")
        return w, p, b, q

    def step(self, winners, players, boards, poses):
        """one SGD step; returns the loss before the step"""
        w, p, b, q = self._batch(winners, players, boards, poses)
        loss = C.c_float()
        _chk(lib().c6_trainer_step(self.h, _p(w, C.c_float), _p(p, C.c_float), _p(b, C.c_float), _p(q, C.c_float), 0, C.byref(loss)))
        return loss.value

    def loss(self, winners, players, boards, poses):
        w, p, b, q = self._batch(winners, players, boards, poses)
        loss = C.c_float()
        _chk(lib().c6_trainer_loss(self.h, _p(w, C.c_float), _p(p, C.c_float), _p(b, C.c_float), _p(q, C.c_float), 0, C.byref(loss)))
        return loss.value

    def step_replay(self, replay, seed=0, draw_id=0):
        """buffer.sample() + step without leaving the device"""
        loss = C.c_float()
        _chk(lib().c6_trainer_step_replay(self.h, replay.h, seed, draw_id, C.byref(loss)))
        return loss.value


class DefaultPolicy:
    """policy/default_policy/mod.rs:84-275, pure MCTS with random playouts, for `num_games` independent policies on the GPU
    (include/connect6_amd.h part 5).  next(boards) is Policy::next for every slot at once; with num_games == 1 the object is
    also a `play()` policy (board[N,N] -> (row, col))."""

    def __init__(self, board_size=15, num_games=1, num_iter=50, seed=0, max_nodes=0, device=0):
        self.n, self.G, self.num_iter = board_size, num_games, num_iter
        h = C.c_void_p()
        _chk(lib().c6_rollout_create(board_size, num_games, num_iter, seed, max_nodes, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().c6_rollout_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, mask=None):
        """DefaultPolicy::new for the masked slots (all by default)"""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _chk(lib().c6_rollout_reset(self.h, None if m is None else _p(m, C.c_uint8)))

    def next(self, boards, active=None):
        """boards int8 [G, N*N] (or [G, N, N]) -> int32 [G] cells (row * N + col), -1 for inactive slots / full boards"""
        b = np.ascontiguousarray(boards, np.int8).reshape(self.G, self.n * self.n)
        a = None if active is None else np.ascontiguousarray(active, np.uint8)
        pos = np.empty(self.G, np.int32)
        _chk(lib().c6_rollout_next(self.h, _p(b, C.c_int8), None if a is None else _p(a, C.c_uint8), _p(pos, C.c_int32), 0))
        return pos

    def __call__(self, board):
        if self.G != 1:
            raise C6Error("DefaultPolicy as a play() policy needs num_games == 1")
        b = np.rint(np.asarray(board)).astype(np.int8)
        c = int(self.next(b.reshape(1, -1))[0])
        return None if c < 0 else (c // self.n, c % self.n)

    def info(self, g=0):
        i = RolloutInfo()
        _chk(lib().c6_rollout_slot_info(self.h, g, C.byref(i)))
        return dict(nodes=i.nodes, edges=i.edges, searches=i.searches, uid=i.uid)

    def node(self, g, board, cap=1024):
        """None, or (visit, black_win, [(cell, visit, black_win) of the children in push order])"""
        b = np.ascontiguousarray(board, np.int8).reshape(-1)
        f, v, w, k = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        cc, cv, cw = (np.zeros(cap, np.int32) for _ in range(3))
        _chk(lib().c6_rollout_node(self.h, g, _p(b, C.c_int8), C.byref(f), C.byref(v), C.byref(w), C.byref(k), cap, _p(cc, C.c_int32),
                                   _p(cv, C.c_int32), _p(cw, C.c_int32)))
        if not f.value:
            return None
        m = min(k.value, cap)
        return v.value, w.value, [(int(cc[i]), int(cv[i]), int(cw[i])) for i in range(m)]

    def bench_playouts(self, boards, iters=5):
        """(ms per launch, stones placed per launch, winners int8 [G]) of uniformly random playouts from `boards`"""
        b = np.ascontiguousarray(boards, np.int8).reshape(self.G, self.n * self.n)
        ms, steps = C.c_double(), C.c_double()
        win = np.zeros(self.G, np.int8)
        _chk(lib().c6_rollout_bench_playouts(self.h, _p(b, C.c_int8), iters, C.byref(ms), C.byref(steps), _p(win, C.c_int8)))
        return ms.value, steps.value, win


def replay_moves(n, moves):
    """Host replay of a move list: [(turn, board_before, cell)], Agent::play bookkeeping (agent_impl/mod.rs:139-145)"""
    board = np.zeros(n * n, np.int8)
    turn, remain = BLACK, 1
    out = []
    for m in moves:
        out.append((turn, board.reshape(n, n).copy(), int(m)))
        board[int(m)] = turn
        remain -= 1
        if remain <= 0:
            remain, turn = 2, -turn
    return out


# ---------------------------------------------------------------------------------------------
# stand-alone batched operators (host arrays in/out)
# ---------------------------------------------------------------------------------------------
def to_soa(boards):
    """[G, N, N] or [G, N*N] int8 -> SoA [N*N, G]"""
    b = np.asarray(boards, dtype=np.int8)
    return np.ascontiguousarray(b.reshape(b.shape[0], -1).T)


def board_step(n, boards_soa, turn, remain, pos):
    b = np.ascontiguousarray(boards_soa, dtype=np.int8).copy()
    G = b.shape[1]
    t = np.ascontiguousarray(turn, dtype=np.int8).copy()
    r = np.ascontiguousarray(remain, dtype=np.int8).copy()
    p = np.ascontiguousarray(pos, dtype=np.int32)
    w = np.zeros(G, np.int8)
    e = np.zeros(G, np.int8)
    _chk(lib().c6_board_step(n, G, _p(b, C.c_int8), _p(t, C.c_int8), _p(r, C.c_int8), _p(p, C.c_int32), _p(w, C.c_int8), _p(e, C.c_int8)))
    return b, t, r, w, e


def search_winner(n, boards_soa):
    b = np.ascontiguousarray(boards_soa, dtype=np.int8)
    w = np.zeros(b.shape[1], np.int8)
    _chk(lib().c6_search_winner(n, b.shape[1], _p(b, C.c_int8), _p(w, C.c_int8)))
    return w


def legal_mask(n, boards_soa):
    b = np.ascontiguousarray(boards_soa, dtype=np.int8)
    G = b.shape[1]
    m = np.zeros(((n * n + 7) // 8, G), np.uint8)
    _chk(lib().c6_legal_mask(n, G, _p(b, C.c_int8), _p(m, C.c_uint8)))
    return m


def augment8(n, boards):
    b = np.ascontiguousarray(boards, dtype=np.int8).reshape(-1, n * n)
    out = np.zeros((b.shape[0], 8, n, n), np.int8)
    _chk(lib().c6_augment8(n, b.shape[0], _p(b, C.c_int8), _p(out, C.c_int8)))
    return out


def recover8(n, boards, values8, policies8):
    b = np.ascontiguousarray(boards, dtype=np.int8).reshape(-1, n * n)
    v8 = np.ascontiguousarray(values8, dtype=np.float32).reshape(-1, 8)
    p8 = np.ascontiguousarray(policies8, dtype=np.float32).reshape(-1, 8, n * n)
    val = np.zeros(b.shape[0], np.float32)
    pri = np.zeros((b.shape[0], n, n), np.float32)
    _chk(lib().c6_recover8(n, b.shape[0], _p(b, C.c_int8), _p(v8, C.c_float), _p(p8, C.c_float), _p(val, C.c_float), _p(pri, C.c_float)))
    return val, pri


def net_forward(n, channels, blocks, blob, boards, player, precision=PREC_DEFAULT):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    b = np.ascontiguousarray(boards, dtype=np.int8).reshape(-1, n * n)
    pl = np.ascontiguousarray(np.broadcast_to(np.asarray(player, dtype=np.int8), (b.shape[0],)))
    pol = np.zeros((b.shape[0], n * n), np.float32)
    val = np.zeros(b.shape[0], np.float32)
    _chk(lib().c6_net_forward_prec(n, channels, blocks, precision, _p(blob, C.c_float), blob.size, b.shape[0], _p(b, C.c_int8),
                                   _p(pl, C.c_int8), _p(pol, C.c_float), _p(val, C.c_float)))
    return pol, val


def bench_conv(n, channels, nboards, iters=3, precision=PREC_FP32, with_residual=0):
    """average ms of one 3x3 conv launch of a PREC_* arithmetic on random data (development / profiling hook)"""
    ms = C.c_double()
    _chk(lib().c6_bench_conv(n, channels, nboards, iters, precision, 0, with_residual, C.byref(ms)))
    return ms.value


def blob_floats(n, channels, blocks):
    return int(lib().c6_net_blob_floats(n, channels, blocks))


def _blob_names(blocks):
    """tensor names of WeightedPolicy (AlphaZero/cpp_weighted/model.hpp:10-50) in BLOB order = forward order, each BatchNorm's
    running statistics right behind its affine parameters.  (libtorch's registration order differs: heads before the tower,
    buffers listed separately -- which is why the blob is assembled by name.)"""
    def bn(p):
        return [p + ".weight", p + ".bias", p + ".running_mean", p + ".running_var"]
    names = ["conv.weight", "conv.bias"] + bn("norm")
    for i in range(blocks - 1):
        for j in range(2):
            names += [f"conv{i}_{j}.weight", f"conv{i}_{j}.bias"] + bn(f"norm{i}_{j}")
    names += ["policy_conv.weight", "policy_conv.bias"] + bn("policy_norm") + ["policy_linear.weight", "policy_linear.bias"]
    names += ["value_conv.weight", "value_conv.bias"] + bn("value_norm")
    names += ["value_linear1.weight", "value_linear1.bias", "value_linear2.weight", "value_linear2.bias"]
    return names


def blob_from_state_dict(sd, n, channels, blocks, header=True):
    """Weight blob for Engine.load_weights / Trainer.load_weights from a state_dict of the reference's WeightedPolicy
    (model.hpp:10-50; names as its register_module calls: conv, norm, conv{i}_{j}, norm{i}_{j}, policy_*, value_*).
    `sd`: mapping name -> array-like (torch tensors, numpy arrays; a torch.load()-ed / torch.jit-exported state_dict works as it
    is).  Tensors are looked up BY NAME and shape-checked, so a re-ordered state_dict cannot produce a permuted blob.
    Aliases: ModuleList-style names tower.{i}.{0,1,2,3}.* (tests/torch_ref.py) are accepted too.
    header=True prepends the 8-float self-describing header (geometry + checksum) that load_weights verifies."""
    cells = n * n
    C_ = channels
    shapes = {"conv.weight": (C_, 2, 3, 3), "policy_conv.weight": (2, C_, 1, 1), "policy_linear.weight": (cells, 2 * cells),
              "value_conv.weight": (1, C_, 1, 1), "value_linear1.weight": (256, cells), "value_linear2.weight": (1, 256),
              "conv.bias": (C_,), "policy_conv.bias": (2,), "policy_linear.bias": (cells,), "value_conv.bias": (1,),
              "value_linear1.bias": (256,), "value_linear2.bias": (1,)}
    for p_, k in (("norm", C_), ("policy_norm", 2), ("value_norm", cells)):
        for f in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{p_}.{f}"] = (k,)
    for i in range(blocks - 1):
        for j in range(2):
            shapes[f"conv{i}_{j}.weight"] = (C_, C_, 3, 3)
            shapes[f"conv{i}_{j}.bias"] = (C_,)
            for f in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"norm{i}_{j}.{f}"] = (C_,)

    def alias(name):
        import re
        m = re.match(r"(conv|norm)(\d+)_(\d+)\.(.*)", name)
        if m:
            kind, i, j, f = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
            return f"tower.{i}.{2 * j + (0 if kind == 'conv' else 1)}.{f}"
        return None

    parts = []
    for name in _blob_names(blocks):
        t = sd.get(name) if hasattr(sd, "get") else None
        if t is None and alias(name) is not None:
            t = sd.get(alias(name))
        if t is None:
            raise C6Error(f"state_dict lacks {name}")
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        if tuple(a.shape) != shapes[name]:
            raise C6Error(f"{name}: shape {tuple(a.shape)}, expected {shapes[name]}")
        parts.append(np.ascontiguousarray(a, dtype=np.float32).reshape(-1))
    payload = np.concatenate(parts)
    assert payload.size == blob_floats(n, channels, blocks)
    return add_blob_header(payload, n, channels, blocks) if header else payload


def add_blob_header(payload, n, channels, blocks):
    """payload (blob_floats floats) -> self-describing blob: 8-float header (magic, version, N, C, blocks, size, checksum) + payload"""
    H = int(lib().c6_net_blob_header_floats())
    out = np.empty(H + payload.size, np.float32)
    out[H:] = np.asarray(payload, np.float32).reshape(-1)
    _chk(lib().c6_net_blob_write_header(_p(out, C.c_float), out.size, n, channels, blocks))
    return out


def check_blob(blob, n, channels, blocks):
    """verify a headed blob (raises C6Error with the reason); returns the payload view"""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    _chk(lib().c6_net_blob_check_header(_p(blob, C.c_float), blob.size, n, channels, blocks))
    return blob[int(lib().c6_net_blob_header_floats()):]


def save_blob(path, blob_or_sd, n, channels, blocks):
    """write a self-describing weight blob (.c6w: raw little-endian f32, header first).  blob_or_sd: a state_dict, a payload
    or an already headed blob"""
    if hasattr(blob_or_sd, "keys"):
        b = blob_from_state_dict(blob_or_sd, n, channels, blocks, header=True)
    else:
        b = np.ascontiguousarray(blob_or_sd, dtype=np.float32).reshape(-1)
        if b.size == blob_floats(n, channels, blocks):
            b = add_blob_header(b, n, channels, blocks)
        else:
            check_blob(b, n, channels, blocks)
    b.astype("<f4").tofile(path)
    return b


def load_blob(path, n, channels, blocks):
    """read a .c6w file written by save_blob, verify header and checksum, return the headed blob"""
    b = np.fromfile(path, dtype="<f4")
    check_blob(b, n, channels, blocks)
    return b


def conv_pack_m8(channels, w_oihw):
    """host-side weight image of the mixed-precision convolution (no GPU needed): uint8 array"""
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    nbytes = lib().c6_conv_pack_m8_bytes(channels)
    out = np.zeros(nbytes, np.uint8)
    _chk(lib().c6_conv_pack_m8(channels, _p(w, C.c_float), _p(out, C.c_uint8), nbytes))
    return out


def stress_eval(n, player, boards):
    """C6_EVAL_STRESS's peaked evaluator on the host, bit for bit (include/connect6_amd.h): boards [len, N*N] -> (values, priors)"""
    b = np.asarray(boards).reshape(-1, n * n)
    rr, cc = np.divmod(np.arange(n * n), n)
    d2 = (rr - n // 2) ** 2 + (cc - n // 2) ** 2
    pol = np.zeros(b.shape, np.float32)
    for i in range(b.shape[0]):
        empty = b[i] == 0
        if empty.any():
            k = np.clip(d2 - d2[empty].min(), 0, 6)
            pol[i] = np.where(empty, np.ldexp(np.float32(1.0), (-20 * k).astype(np.int32)), np.float32(0.0))
    return np.full(b.shape[0], -0.875 * player, np.float32), pol


def conv_wgrad(n, channels, dy, x, method):
    """dW [C, C, 3, 3] of a 3x3 convolution from dy and x (NHWC [B, N*N, C], f32): method 0 direct kernel, 1 Winograd domain (C = 256)"""
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    B = dy.shape[0]
    dw = np.zeros((channels, channels, 3, 3), np.float32)
    _chk(lib().c6_conv_wgrad_op(n, channels, B, _p(dy, C.c_float), _p(x, C.c_float), method, _p(dw, C.c_float)))
    return dw


def bench_board(n, num_games, iters=20, scan_only=False):
    """average ms of one K1 launch over `num_games` resident games: a board-step (one stone per game) or, scan_only,
    the search-only launch (is_game_end of every game)"""
    ms = C.c_double()
    _chk((lib().c6_bench_board_scan if scan_only else lib().c6_bench_board)(n, num_games, iters, C.byref(ms)))
    return ms.value


def dirichlet_noise(n, alpha, k0, k1, search, depth, boards):
    b = np.ascontiguousarray(boards, dtype=np.int8).reshape(-1, n * n)
    num = b.shape[0]
    a = [np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.uint32), (num,))) for x in (k0, k1, search, depth)]
    out = np.zeros((num, n, n), np.float32)
    _chk(lib().c6_dirichlet_noise(n, num, alpha, _p(a[0], C.c_uint32), _p(a[1], C.c_uint32), _p(a[2], C.c_uint32),
                                  _p(a[3], C.c_uint32), _p(b, C.c_int8), _p(out, C.c_float)))
    return out
