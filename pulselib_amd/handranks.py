"""HandRanks table management (replaces the manual download at
/root/reference environments/Poker/PokerGPU.py:47-58).

The 2+2 table (int32[32,487,834], 130 MB) is built by the native generator
(`pulse_handranks_generate`, csrc/handranks_gen.cpp), cached on disk as a byte-compatible
`HandRanks.dat`, and kept ONCE per device in HBM however many environments are created.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np
import torch

from . import _native

_PKG = Path(__file__).resolve().parent
_DEVICE_TABLES: dict[str, torch.Tensor] = {}
_HOST_TABLE: np.ndarray | None = None


def default_path() -> Path:
    env = os.environ.get("PULSE_HANDRANKS")
    if env:
        return Path(env)
    return _PKG / "environments" / "Poker" / "HandRanks.dat"


def generate(n_threads: int = 0) -> np.ndarray:
    out = np.empty(_native.HANDRANKS_LEN, dtype=np.int32)
    _native.check(_native.lib().pulse_handranks_generate(out.ctypes.data_as(C.c_void_p), n_threads), "pulse_handranks_generate")
    return out


def host_table(path: Path | None = None) -> np.ndarray:
    """int32 numpy table: read from the cache file if it has the right size, else generate + cache."""
    global _HOST_TABLE
    if _HOST_TABLE is not None:
        return _HOST_TABLE
    path = Path(path) if path else default_path()
    if path.exists() and path.stat().st_size == _native.HANDRANKS_LEN * 4:
        _HOST_TABLE = np.fromfile(path, dtype=np.int32)
        return _HOST_TABLE
    table = generate()
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_name(path.name + f".tmp{os.getpid()}")
        table.tofile(tmp)
        os.replace(tmp, path)
    except OSError:
        pass   # read-only tree: keep the in-memory copy
    _HOST_TABLE = table
    return table


def device_table(device) -> torch.Tensor:
    device = torch.device(device)
    key = str(device) if device.index is not None else f"{device.type}:{torch.cuda.current_device()}"
    t = _DEVICE_TABLES.get(key)
    if t is None:
        t = torch.from_numpy(host_table()).to(device)
        _DEVICE_TABLES[key] = t
    return t
