"""Where the 16-row training kernel (csrc/qnet_train16.h) spends its time: the diagnostic build stores the clock of thread 0 at
the phase boundaries of each workgroup's (last) tile.  Read the SHARES.
    PULSE_TRAIN_TILE=16 python tools/qnet_train16_timeline.py [n_rows] [mask fraction]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pulselib_amd import _native  # noqa: E402

_native._SO = Path(os.environ.get("PULSE_STAMPS_LIB", ROOT / "pulselib_amd" / "libpulse_hip_stamps.so"))
os.environ.setdefault("PULSE_TRAIN_TILE", "16")
from pulselib_amd.environments.Poker import PokerQNetwork  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
dev = torch.device("cuda:0")
lib = _native.lib()
lib.pulse_debug_set_qnet_stamp_buffer.argtypes = [C.c_void_p]
lib.pulse_debug_set_qnet_stamp_buffer.restype = C.c_int
q = PokerQNetwork(None, dev, gamma=.95, update_freq=20, state_dim=40)
g = torch.Generator(device="cpu").manual_seed(0)
s = torch.randn((N, 40), generator=g).to(dev); s[:, 12] = 0
ns = torch.randn((N, 40), generator=g).to(dev)
a = torch.randint(0, 13, (N,), generator=g).to(dev)
r = torch.randn((N,), generator=g).to(dev)
d = (torch.rand((N,), generator=g) < 0.3).to(dev)
m = (torch.rand((N,), generator=g) < frac).to(dev)
buf = torch.zeros((1024, 16), dtype=torch.int64, device=dev)
ORDER = [0, 1, 10, 11, 12, 13, 2, 3, 4, 5, 6, 7, 8, 9]
names = ["row lists, bookkeeping, (earlier tiles)", "gather + layer 1 (both networks)", "layer 2", "layer 3", "layer 4", "layer 5 + max Q_target",
         "delta_5", "layer 5 bwd", "layer 4 bwd", "layer 3 bwd", "layer 2 bwd", "layer 1 bwd", "bias + stats store"]
for rep in range(3):
    q.train_step_native(s, a, r, ns, d, m)
torch.cuda.synchronize()
acc = np.zeros(len(ORDER) - 1)
for rep in range(5):
    buf.zero_()
    lib.pulse_debug_set_qnet_stamp_buffer(buf.data_ptr())
    q.train_step_native(s, a, r, ns, d, m)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    st = st[(st[:, 9] > 0) & (st[:, 1] > 0)]
    acc += np.diff(st[:, ORDER], axis=1).mean(axis=0)
    span = st[:, 9].max() - st[:, 0].min()
lib.pulse_debug_set_qnet_stamp_buffer(None)
acc /= 5
print(f"N={N} mask fraction {frac}: {len(st)} workgroups with a tile, kernel span {span} ticks; last tile of a workgroup: {acc[1:-1].sum():.0f} ticks")
for n, c in zip(names, acc):
    print(f"   {n:42s} {c:9.0f}  {100 * c / acc.sum():5.1f} %")
