"""The C-ABI library loads without a GPU and exports every symbol include/pulse_env.h declares; argument
validation works on the host side (no compute call is made here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (ROOT / "include" / "pulse_env.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pulse_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pulselib_amd import _native
    lib = _native.lib()
    names = _declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"libpulse_hip.so does not export {name}"
        assert name in _native.SYMBOLS, f"{name} has no ctypes signature in pulselib_amd/_native.py"
    assert sorted(_native.SYMBOLS) == names
    assert lib.pulse_version() == 1


def test_error_reporting_and_validation_without_gpu():
    from pulselib_amd import _native
    lib = _native.lib()
    assert lib.pulse_handranks_generate(None, 1) == -1
    assert b"out is null" in lib.pulse_last_error()
    v = _native.PokerView()
    v.n_games, v.n_players, v.active_players, v.max_players, v.obs_size = 4, 20, 20, 20, 70
    assert lib.pulse_poker_step(C.byref(v), None, None, None) == -1
    assert b"unsupported shape" in lib.pulse_last_error()
    with pytest.raises(ValueError):
        _native.check(-1, "probe")
    # the learner's entry points check their scratch before anything is launched
    net = _native.QNet()
    net.state_dim, net.n_actions = 40, 13
    dummy = (C.c_int32 * 8)()
    ptr = C.addressof(dummy)
    assert lib.pulse_qnet_act_select(C.byref(net), ptr, 40, 1024, ptr, 0, 0.1, 1, 1, 0, ptr, None, ptr, None, 0, None) == -1
    assert b"pulse_qnet_act_select: needs" in lib.pulse_last_error()
    assert lib.pulse_qnet_act_select(C.byref(net), ptr, 40, 1024, ptr, 0, 0.1, 1, 1, 0, ptr, None, ptr, ptr, 100, None) == -1
    assert b"259 words per 256 rows" in lib.pulse_last_error()
    net.state_dim = 100                                    # the cooperative tile takes up to 64 inputs
    assert lib.pulse_qnet_act_select(C.byref(net), ptr, 100, 1024, ptr, 0, 0.1, 1, 1, 0, ptr, None, ptr, ptr, 1 << 20, None) == -1


def test_product_classes_refuse_cpu_devices():
    import torch
    from pulselib_amd.environments.Poker import PokerGPU
    from pulselib_amd.environments.blackjack import BlackJack
    from pulselib_amd.environments.Particle2D import Particle2D
    from pulselib_amd.environments.TFE import TFEBatch
    for make in (lambda: PokerGPU(torch.device("cpu"), []), lambda: BlackJack(torch.device("cpu"), 4),
                 lambda: Particle2D(torch.device("cpu"), 4), lambda: TFEBatch(torch.device("cpu"), 4)):
        with pytest.raises(RuntimeError, match="No CPU fallback|no CPU fallback"):
            make()


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pulselib_amd/ may import, include or link it."""
    for path in (ROOT / "pulselib_amd").rglob("*.py"):
        src = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{path} imports oracle/"
        assert "liboracle" not in src, f"{path} loads liboracle.so"
    for path in (ROOT / "pulselib_amd" / "csrc").glob("*"):
        if path.is_file():
            src = path.read_text()
            assert not re.search(r'#include\s+"[^"]*oracle', src), f"{path} includes oracle sources"
            assert "liboracle" not in src and "-loracle" not in src, f"{path} links the oracle"
