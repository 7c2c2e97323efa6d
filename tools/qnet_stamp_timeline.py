"""Where the training kernel (csrc/qnet.hip: qnet_train8_kernel, or qnet_train_kernel with PULSE_TRAIN_WAVES=4) spends its time: runs the diagnostic build
(libpulse_hip_stamps.so, `make -C pulselib_amd/csrc stamps`), whose kernel stores the clock at the phase boundaries
of each workgroup's tile, and prints mean ticks per segment.  Read the SHARES."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pulselib_amd import _native  # noqa: E402

import os  # noqa: E402
_native._SO = Path(os.environ.get("PULSE_STAMPS_LIB", ROOT / "pulselib_amd" / "libpulse_hip_stamps.so"))
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
buf = torch.zeros((256, 16), dtype=torch.int64, device=dev)
names = ["compaction", "both forwards", "delta_5", "layer 5 bwd", "layer 4 bwd", "layer 3 bwd", "layer 2 bwd", "layer 1 bwd",
         "bias + stats store"]
for rep in range(3):
    q.train_step_native(s, a, r, ns, d, m)
torch.cuda.synchronize()
acc = np.zeros(9)
for rep in range(5):
    buf.zero_()
    lib.pulse_debug_set_qnet_stamp_buffer(buf.data_ptr())
    q.train_step_native(s, a, r, ns, d, m)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    st = st[st[:, 9] > 0]
    acc += np.diff(st[:, :10], axis=1).mean(axis=0)
    extra = (st[:, 10] - st[:, 6]).mean(), (st[:, 11] - st[:, 10]).mean(), (st[:, 7] - st[:, 11]).mean()
    span = st[:, 9].max() - st[:, 0].min()
lib.pulse_debug_set_qnet_stamp_buffer(None)
acc /= 5
print(f"N={N} mask fraction {frac}: kernel span {span} ticks; per workgroup (first... last tile overwrite): total {acc.sum():.0f}")
if (st[:, 10] > 0).all():          # the four-wavefront kernel (PULSE_TRAIN_WAVES=4) stamps inside the layer-2 phase
    print(f"   layer 2 bwd split: dW blocks {extra[0]:.0f}, delta_1 tile {extra[1]:.0f}, wait at barrier {extra[2]:.0f}")
for n, c in zip(names, acc):
    print(f"   {n:22s} {c:9.0f}  {100 * c / acc.sum():5.1f} %")
