"""ctypes binding of libpulse_hip.so (include/pulse_env.h).

The library is the product: if it is missing or does not load, importing this module's `lib()`
raises -- there is no PyTorch/CPU fallback for any environment step.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_SO = Path(os.environ["PULSE_LIB"]) if os.environ.get("PULSE_LIB") else _PKG / "libpulse_hip.so"     # PULSE_LIB: A/B of builds (tools/ab_bench.py)
_CSRC = _PKG / "csrc"
_LIB: C.CDLL | None = None

HANDRANKS_LEN = 32487834
STATS_SLOTS, STATS_STRIDE = 256, 16          # PulsePokerResetOpts.stats_out layout (include/pulse_env.h)
MAX_SEATS = 16

# phase bits (include/pulse_env.h)
PH_CAPTURE, PH_EQUITY, PH_EXECUTE, PH_ADVANCE = 0x001, 0x002, 0x004, 0x008
PH_FOLDWIN, PH_SHOWDOWN, PH_CLEARDONE, PH_REWARD, PH_OBS = 0x010, 0x020, 0x040, 0x080, 0x100
PH_STEP = 0x1FF

# PulsePokerView.flags: kernel variants selectable per view (include/pulse_env.h)
VIEW_NO_OBS_STAGING, VIEW_NO_CHUNK, VIEW_FOUR_LANES, VIEW_NO_PAIRS = 0x1, 0x2, 0x4, 0x8

AGENT_EXTERNAL, AGENT_RANDOM, AGENT_HEURISTIC_HANDS, AGENT_TIGHT_AGGRESSIVE, AGENT_LOOSE_PASSIVE, AGENT_SMALL_BALL = range(6)


class PulseError(RuntimeError):
    pass


class PokerView(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("n_games", "n_players", "active_players", "max_players", "obs_size", "hand_ranks_len",
                                  "flags", "reserved0")]
        + [(n, C.c_void_p) for n in (
            "hand_ranks",
            "pots", "stages", "deck_positions", "button", "sb", "bb", "idx", "highest", "agg", "acted",
            "last_raise_size", "prev_stacks", "prev_invested",
            "is_done", "is_done_out", "equity_dirty",
            "stacks", "current_round_bet", "total_invested", "status",
            "hands", "board", "decks", "equities", "obs",
            "w1", "w2", "K", "alpha",
            "pre_board", "pre_hands", "pre_eq", "pre_rank")]
    )


class PokerResetOpts(C.Structure):
    _fields_ = [("first", C.c_int32), ("starting_bbs", C.c_int32), ("max_bbs", C.c_int32), ("rotation", C.c_int32),
                ("seed", C.c_uint64), ("episode", C.c_uint64), ("table_id0", C.c_uint64),
                ("prefixed_decks", C.c_void_p), ("decks_out", C.c_void_p),
                ("shuffle_key_bits", C.c_int32), ("reserved0", C.c_int32), ("stats_rewards", C.c_void_p), ("stats_out", C.c_void_p)]


class QNet(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("n_actions", C.c_int32)] + [(n, C.c_void_p) for n in (
        "w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "w5", "b5")]


class QNetTrain(C.Structure):
    _fields_ = [("net", QNet), ("target", QNet)] + [(n, C.c_void_p) for n in (
        "params", "target_params", "grad", "exp_avg", "exp_avg_sq", "step", "stats", "report", "partials")] + [(n, C.c_float) for n in (
            "lr", "weight_decay", "beta1", "beta2", "eps", "max_grad_norm", "gamma", "dropout_p")] + [
        ("update_freq", C.c_int32), ("max_blocks", C.c_int32), ("select_scratch", C.c_void_p), ("select_words", C.c_int64), ("select_from_act", C.c_int32),
        ("separate_apply", C.c_int32), ("meet_wait_ticks", C.c_int64), ("debug_meet_extra", C.c_int32), ("reserved0", C.c_int32)]


class QNetAct(C.Structure):
    _fields_ = [("states", C.c_void_p), ("row_stride", C.c_int64), ("seat_idx", C.c_void_p), ("q_seat", C.c_int32), ("epsilon", C.c_float),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("table_id0", C.c_uint64), ("terminated", C.c_void_p), ("row_mask_out", C.c_void_p),
                ("select_scratch", C.c_void_p), ("select_words", C.c_int64)]


class QTable(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("capacity", C.c_uint64), ("region_slots", C.c_uint64)]


class QTableScratch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("count", "cells", "targets", "owner", "acc_key", "acc_cnt", "acc_sum")] + [
        ("n", C.c_uint32), ("acc_slots", C.c_uint32), ("wait_ticks", C.c_int64), ("debug_meet_extra", C.c_int32), ("reserved0", C.c_int32)]


class BlackjackView(C.Structure):
    _fields_ = [("batch_size", C.c_int32)] + [(n, C.c_void_p) for n in (
        "decks", "deck_positions", "players_cards", "players_card_idx", "player_card_sums",
        "dealer_cards", "dealer_card_idx", "dealer_upcard", "dealer_card_sums",
        "terminated", "has_ace", "dealer_has_ace", "rewards", "obs")]


# every symbol include/pulse_env.h declares: (restype, argtypes)
_P, _I32, _U32, _U64, _F32, _I64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float, C.c_int64
SYMBOLS = {
    "pulse_version": (C.c_int, []),
    "pulse_last_error": (C.c_char_p, []),
    "pulse_handranks_generate": (C.c_int, [_P, C.c_int]),
    "pulse_poker_eval_hands": (C.c_int, [_P, _I32, _P, _I32, _I32, _I32, _P, _P]),
    "pulse_poker_eval_closed_form": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "pulse_poker_step": (C.c_int, [_P, _P, _P, _P]),
    "pulse_poker_phases": (C.c_int, [_P, _U32, _P, _P, _P, _P]),
    "pulse_poker_reset": (C.c_int, [_P, _P, _P]),
    "pulse_poker_policy": (C.c_int, [_P, _I32, _P, _I32, _P, _I32, _U64, _U64, _U64, _P, _P]),
    "pulse_poker_policy_step": (C.c_int, [_P, _P, _U64, _U64, _U64, _P, _P, _P]),
    "pulse_poker_rollout": (C.c_int, [_P, _P, _P, _U64, _U64, _U64, _P, _P, _P, _I32, _P, _P, _P]),
    "pulse_poker_rollout_until": (C.c_int, [_P, _P, _P, _U64, _U64, _U64, _P, _P, _P, _I32, _I32, _P, _I32, _P, _P, _P, _P]),
    "pulse_timer_create": (C.c_int, [_P]),
    "pulse_timer_collect": (C.c_int, [_P, _P, _P, _P]),
    "pulse_timer_destroy": (C.c_int, [_P]),
    "pulse_stoprule_create": (C.c_int, [_I32, _I64, C.c_double, _I32, _P, C.c_char_p, _I32, _I32, _P]),
    "pulse_stoprule_submit": (C.c_int, [_P, _P, _I32, _P]),
    "pulse_stoprule_counts": (C.c_int, [_P, _P, _P, _P]),
    "pulse_stoprule_decide": (C.c_int, [_P, _P]),
    "pulse_stoprule_drain": (C.c_int, [_P]),
    "pulse_stoprule_publish": (C.c_int, [_P]),
    "pulse_stoprule_destroy": (C.c_int, [_P]),
    "pulse_stoprule_mode": (C.c_int, [_P]),
    "pulse_stoprule_side_launches": (_I64, [_P]),
    "pulse_stoprule_set_option": (C.c_int, [_P, _I32, _I64]),
    "pulse_stoprule_stats": (C.c_int, [_P, _P]),
    "pulse_shm_create": (C.c_int, [C.c_char_p, _I32, _I32, _P]),
    "pulse_shm_all_sum": (C.c_int, [_P, _I64, _I64, _P]),
    "pulse_shm_set_device": (C.c_int, [_P, C.c_char_p]),
    "pulse_shm_device_is_private": (C.c_int, [_P, _I32]),
    "pulse_shm_destroy": (C.c_int, [_P]),
    "pulse_comm_unique_id": (C.c_int, [_P]),
    "pulse_comm_create": (C.c_int, [_P, _I32, _I32, _P]),
    "pulse_comm_all_reduce_i64": (C.c_int, [_P, _P, _P, _I32, _P]),
    "pulse_comm_destroy": (C.c_int, [_P]),
    "pulse_poker_ablate": (C.c_int, [_P, _U32, _P, _P, _U64, _U64, _P]),
    "pulse_qnet_called_off_meetings": (_I64, []),
    "pulse_poker_act_policy_step": (C.c_int, [_P, _P, _U64, _U64, _U64, _P, _P, _P, _P, _P, _P]),
    "pulse_calib_stream": (C.c_int, [_P, _U64, _I32, _P]),
    "pulse_poker_stats": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P]),
    "pulse_poker_hand_metrics": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _I32, _I32, _I32, _P, _I32, _P, _P, _P]),
    "pulse_blackjack_reset": (C.c_int, [_P, _P, _P, _U64, _U64, _P]),
    "pulse_blackjack_step": (C.c_int, [_P, _P, _P]),
    "pulse_tfe_reset": (C.c_int, [_P, _P, _I32, _I32, _U64, _U64, _P]),
    "pulse_tfe_step": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _U64, _U64, _U64, _P]),
    "pulse_qtable_select": (C.c_int, [_P, _P, _I32, _I32, C.c_double, _U64, _U64, _U64, _P, _P, _P]),
    "pulse_qtable_update": (C.c_int, [_P, _P, _U64, _P, _P, _P, _P, _P, _I32, _I32, C.c_double, C.c_double, _P]),
    "pulse_qtable_rollout_step": (C.c_int, [_P, _P, _U64, _P, _P, _I32, _I32, C.c_double, C.c_double, C.c_double, _U64, _U64, _U64, _U64, _U64,
                                            _P, _P, _P, _P, _P]),
    "pulse_particle2d_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _F32, _I32, _P]),
    "pulse_qnet_forward": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "pulse_qnet_act": (C.c_int, [_P, _P, _I64, _I32, _P, _I32, _F32, _U64, _U64, _U64, _P, _P, _P, _P, _P]),
    "pulse_qnet_act_select": (C.c_int, [_P, _P, _I64, _I32, _P, _I32, _F32, _U64, _U64, _U64, _P, _P, _P, _P, _I64, _P]),
    "pulse_qnet_param_count": (C.c_int, [_I32, _I32]),
    "pulse_qnet_slice_floats": (C.c_int, []),
    "pulse_qnet_train_step": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I64, _P, _P, _I32, _U64, _U64, _U64, _P, _P, _P]),
    "pulse_qnet_train_grads": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I64, _P, _P, _I32, _U64, _U64, _U64, _P, _P, _P]),
    "pulse_qnet_train_apply": (C.c_int, [_P, _P]),
}


def build(force: bool = False) -> Path:
    """Compile libpulse_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = list(_CSRC.glob("*.hip")) + list(_CSRC.glob("*.cpp")) + list(_CSRC.glob("*.h")) + [_PKG.parent / "include" / "pulse_env.h"]
    if force or not _SO.exists() or any(s.stat().st_mtime > _SO.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_CSRC), f"-j{min(8, os.cpu_count() or 1)}"] + (["-B"] if force else []))
    return _SO


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not _SO.exists():
            if os.environ.get("PULSE_NO_AUTOBUILD"):
                raise PulseError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
            build()
        # PyTorch-ROCm ships its own libamdhip64; it must be in the process before this library resolves its HIP
        # dependency, or two HIP runtimes coexist and launches on torch's tensors fail ("no ROCm-capable device")
        import torch  # noqa: F401
        try:
            handle = C.CDLL(str(_SO))
        except OSError as e:  # pragma: no cover - environment specific
            raise PulseError(f"cannot load {_SO}: {e} (the HIP library is required; there is no CPU fallback)") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.pulse_version() != 1:
            raise PulseError("libpulse_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pulse_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise PulseError(f"{what}: {msg} (code {rc})")


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
