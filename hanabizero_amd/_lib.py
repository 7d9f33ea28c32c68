"""ctypes binding of libhanabizero_hip.so (the C ABI declared in include/hz_tree.h and include/hz_env.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the product raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: PyTorch-ROCm brings its own HIP runtime (torch/lib); loaded after this library's, the
#                             process has two and this library's sees no device ("no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HANABIZERO_HIP_LIB") or os.path.join(_HERE, "libhanabizero_hip.so")  # override: diagnostic builds (tools/)


class HzError(RuntimeError):
    pass


class MlpJob(C.Structure):  # include/hz_mlp.h hz_mlp_job_t
    _fields_ = [("ks", C.c_int32), ("src_off", C.c_int32), ("dst_off", C.c_int32), ("res_off", C.c_int32),
                ("bias_off", C.c_int32), ("flags", C.c_int32), ("reserved0", C.c_int32), ("producer", C.c_int32)]


class MlpHeader(C.Structure):  # include/hz_mlp.h hz_mlp_header_t
    _fields_ = [("n_jobs", C.c_int32), ("row_stride", C.c_int32), ("hidden", C.c_int32), ("state_off", C.c_int32),
                ("hidden_off", C.c_int32), ("off_reward", C.c_int32), ("off_value", C.c_int32), ("off_policy", C.c_int32),
                ("support_size", C.c_int32), ("support_min", C.c_int32), ("num_actions", C.c_int32),
                ("action_table_stride", C.c_int32), ("in_width", C.c_int32), ("dtype", C.c_int32),
                ("num_waves", C.c_int32), ("tiles_per_wave", C.c_int32),
                ("logit_split", C.c_int32), ("off_reward2", C.c_int32), ("off_value2", C.c_int32), ("lo_plane", C.c_int32),
                ("kstep_stride", C.c_int64), ("wave_stream_off", C.c_int64 * 16)]


class BnFinish(C.Structure):  # include/hz_train.h hz_bn_finish_t
    _fields_ = [("dst0", C.c_void_p), ("dst1", C.c_void_p), ("scratch", C.c_void_p), ("cols", C.c_int32), ("groups", C.c_int32),
                ("momentum", C.c_float), ("reserved", C.c_int32)]


class RowsJob(C.Structure):  # include/hz_rows.h hz_rows_job_t
    _fields_ = [("slot", C.c_void_p), ("list", C.c_void_p), ("count", C.c_void_p), ("num_arrays", C.c_int32),
                ("max_rows", C.c_int32), ("src", C.c_void_p * 8), ("dst", C.c_void_p * 8), ("row_bytes", C.c_int64 * 8)]


class ActorBufs(C.Structure):  # include/hz_selfplay.h hz_actor_bufs_t
    _fields_ = [(n, C.c_int32) for n in ("num_envs", "num_actions", "packed_words", "max_moves", "outbox_games",
                                         "env_id_base")] + \
               [(n, C.c_void_p) for n in ("action", "reward", "value", "visits", "legal", "obs", "traj_len", "ent_sum",
                                          "meta", "out_action", "out_reward", "out_value", "out_visits", "out_legal",
                                          "out_obs", "out_meta", "out_count", "slot", "finished", "num_finished",
                                          "illegal_steps")]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "hanabizero_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    V, I, F, U64, U32, I64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_int64
    lib.hz_last_error.restype = C.c_char_p
    lib.hz_version.restype = I
    sig = {
        # include/hz_tree.h
        "hz_tree_create": [C.POINTER(V), I, I, I, I],
        "hz_tree_destroy": [V],
        "hz_tree_set_params": [V, I, F, F, F, U64, U32],
        "hz_tree_prepare": [V, F, V, V, V, V, V],
        "hz_tree_traverse": [V, I, V, V, V, V],
        "hz_tree_traverse_gather": [V, I, V, V, V, V, I, I, V, I, I, V],
        "hz_tree_backprop_traverse": [V, I, V, V, V, I, V, V, V, V],
        "hz_tree_backprop_nets": [V, I, V, I64, V, I64, I, I, V, I64, I, V, V, V],
        "hz_tree_backprop": [V, I, V, V, V, V],
        "hz_support_to_scalar": [V, I64, I, I, I, V, I, V],
        "hz_tree_copy": [V, V, V],
        "hz_tree_get_distributions": [V, V, V],
        "hz_tree_get_values": [V, V, V],
        "hz_tree_get_root_stats": [V, V, V, V],
        "hz_tree_get_trajectories": [V, V, I, V],
        "hz_tree_get_minmax": [V, V, V, V],
        "hz_tree_get_root_priors": [V, V, V],
        "hz_tree_get_path_len": [V, V, V],
        # include/hz_env.h
        "hz_env_create": [C.POINTER(V), I, I, I, I, I, I, I, V, I],
        "hz_env_destroy": [V],
        "hz_env_dims": [V, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I)],
        "hz_env_reset": [V, V, V],
        "hz_env_reset_rows": [V, V, C.POINTER(RowsJob), V],
        "hz_env_step": [V, V, V, V, V, V, V, V],
        "hz_env_observe": [V, I, V, I, I64, V, V, V],
        "hz_env_probe": [V, V, V],
        "hz_env_snapshot": [V, V, V, V],
        "hz_env_restore": [V, V, V, V],
        # include/hz_selfplay.h
        "hz_select_action": [I, I, V, V, V, F, I, V, V, V],
        "hz_rows_scatter": [V, V, I64, V, I, V],
        "hz_actor_draw": [U64, I64, V, I, I, C.c_double, V, V, V],
        "hz_actor_record_search": [C.POINTER(ActorBufs), V, V, V, V, F, I, V, V, V],
        "hz_actor_record_step": [C.POINTER(ActorBufs), V, V, V, V, V, V, V],
        "hz_actor_flush": [C.POINTER(ActorBufs), V],
        "hz_actor_flush_job": [C.POINTER(ActorBufs), C.POINTER(RowsJob)],
        "hz_actor_pack": [C.POINTER(ActorBufs), I64, I, I64, V, V, I64, V],
        "hz_actor_begin_move": [C.POINTER(ActorBufs), V, V, V, V, I64, V, I64, I, I64, V],
        "hz_actor_begin_move_draw": [C.POINTER(ActorBufs), V, V, V, V, I64, V, I64, I, I64, U64, V, C.c_double, V, V, V],
        # include/hz_movetail.h
        "hz_actor_move_tail": [V, V, C.POINTER(ActorBufs), I, V, V, V, V, F, V, I, V, V, V, V, V, V, V, V, I64, I, I64, I, U64, V,
                               C.c_double, V, V, V],
        # include/hz_mlp.h
        "hz_mlp_recurrent": [C.POINTER(MlpHeader), V, V, V, V, V, I64, V, I64, V, V, V, V, V, I, I, V],
        "hz_mlp_recurrent_res": [C.POINTER(MlpHeader), V, V, V, V, V, I64, V, I64, V, V, V, V, V, I, I, V, I64, V],
        # include/hz_search.h
        "hz_search_run": [V, I, C.POINTER(MlpHeader), V, V, V, V, V, I64, I64, V, V, V, V, V, V, I, V],
        "hz_search_set_predicted_lines": [V, I],
        "hz_search_poll_giveups": [C.POINTER(C.c_uint)],
        "hz_mlp_poll_giveups": [C.POINTER(C.c_uint)],
        "hz_mlp_poll_giveups_async": [V, V],
        "hz_search_poll_giveups_async": [V, V],
        # include/hz_replay.h
        "hz_replay_windows": [V, I, V, V, I, I, I, V, I64, I64, I, V],
        "hz_replay_windows_seq": [V, I, V, V, V, V, I, I, I, I, I, V, I64, I64, I, V, I, I64, V, V, V],
        "hz_replay_targets": [V, I, I, I, I, I64, V, V, V, V, V, V, V, V, V, V, V, V, V, V],
        # include/hz_train.h
        "hz_bn_act_forward": [V, I64, V, I64, V, I64, I, I, V, V, V, V, F, F, V, V, I, I, V],
        "hz_bn_act_backward": [V, I64, V, I64, V, I64, V, I64, V, I64, I, I, V, V, V, V, V, I, I, V],
        "hz_bn_act_forward_groups": [V, I64, V, I64, V, I64, I, I, I, V, V, V, V, F, F, V, V, V, I, I, V],
        "hz_bn_act_backward_groups": [V, I64, V, I64, V, I64, V, I64, V, I64, I, I, I, V, V, V, V, V, V, I, I, V],
        "hz_bn_groups_finish": [V, I, I, I, V],
        "hz_state_action_rows": [V, I64, V, I64, I, I, I, V, I64, I, V],
        "hz_scale_rows3": [V, I, V, I, V, I, V, I, I, V, V, V, I, V],
        "hz_muzero_head_losses": [V, I64, V, I64, V, I64, I, I, I, I, I, V, I64, V, I64, V, I64, V, F, F, F, V, V, V, V, V, V],
        "hz_muzero_unrolled_losses": [V, I64, V, I64, V, I64, I, I, I, I, I, I, V, I64, I64, V, I64, I64, V, I64, I64, V, F, F, F, V, V, V, V, V, V],
        # include/hz_netglue.h
        "hz_add_relu": [V, I64, V, I64, I, I, I, V],
        "hz_test_expf": [V, V, I64, V],
        "hz_test_expf_checksum": [V, V],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = I
    lib.hz_actor_packed_bytes.argtypes = [I, I64, I, I, V]
    lib.hz_actor_packed_bytes.restype = I64
    lib.hz_tree_hbm_bytes.argtypes = [V]
    lib.hz_tree_hbm_bytes.restype = I64
    lib.hz_env_hbm_bytes.argtypes = [V]
    lib.hz_env_hbm_bytes.restype = I64
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        raise HzError("%s failed (%d): %s" % (what, rc, lib.hz_last_error().decode()))


def poll_giveups():
    """include/hz_mlp.h::hz_mlp_poll_giveups: waits on arrival counters that timed out since the library was loaded (must be 0)."""
    n = C.c_uint(0)
    check(lib.hz_mlp_poll_giveups(C.byref(n)), "hz_mlp_poll_giveups")
    return int(n.value)


def declared_symbols():
    """Names every header under include/ declares (used by the CPU-side ABI test)."""
    import re
    names = []
    inc = os.path.join(os.path.dirname(_HERE), "include")
    for h in sorted(os.listdir(inc)):
        if not h.endswith(".h"):
            continue
        text = open(os.path.join(inc, h)).read()
        names += re.findall(r"\b(hz_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n != "hz_tiebreak_rand" and not n.endswith("_t")))  # types are not symbols
