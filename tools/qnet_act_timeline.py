"""Where the masked action-selection kernel (csrc/qnet.hip: qnet_act4_kernel) spends its time: the diagnostic build
(`make -C pulselib_amd/csrc stamps`) stores the clock at phase boundaries of each workgroup's first wavefront; prints the
mean ticks per segment over the workgroups (the last tile's stamps overwrite earlier tiles').  Read the SHARES."""
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
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1 / 6
dev = torch.device("cuda:0")
lib = _native.lib()
lib.pulse_debug_set_qnet_stamp_buffer.argtypes = [C.c_void_p]
lib.pulse_debug_set_qnet_stamp_buffer.restype = C.c_int
q = PokerQNetwork(None, dev, gamma=.95, update_freq=20, state_dim=40)
g = torch.Generator(device="cpu").manual_seed(0)
s = torch.randn((N, 40), generator=g).to(dev)
seat = (torch.rand((N,), generator=g) < frac).to(torch.int32).to(dev)       # seat 1 = the learner's
acts = torch.zeros(N, dtype=torch.long, device=dev)
mask = torch.zeros(N, dtype=torch.uint8, device=dev)
n_blocks = (N + 127) // 128
buf = torch.zeros((n_blocks, 16), dtype=torch.int64, device=dev)
names = ["weight loads issued + compaction (incl. the training row lists)", "row gather + the five layers + argmax + store", "exit"]
COLS = [0, 1, 7, 8]                     # the stamps the kernel takes (csrc/qnet.hip: qnet_act4_kernel)


net = q._net_struct(q.network)
for rep in range(3):
    _native.check(lib.pulse_qnet_act(C.byref(net), s.data_ptr(), 40, N, seat.data_ptr(), 1, C.c_float(0.1), 1, rep, 0, acts.data_ptr(), None, None,
                                     mask.data_ptr(), torch.cuda.current_stream().cuda_stream), "act")
torch.cuda.synchronize()
acc = np.zeros(3)
for rep in range(5):
    buf.zero_()
    lib.pulse_debug_set_qnet_stamp_buffer(buf.data_ptr())
    _native.check(lib.pulse_qnet_act(C.byref(net), s.data_ptr(), 40, N, seat.data_ptr(), 1, C.c_float(0.1), 1, 10 + rep, 0, acts.data_ptr(), None, None,
                                     mask.data_ptr(), torch.cuda.current_stream().cuda_stream), "act")
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    st = st[st[:, 8] > 0]
    acc += np.diff(st[:, COLS], axis=1).mean(axis=0)
    span = st[:, 8].max() - st[:, 0].min()
    starts = st[:, 0] - st[:, 0].min()
lib.pulse_debug_set_qnet_stamp_buffer(None)
acc /= 5
print(f"N={N} learner fraction {frac:.3f}: kernel span {span} ticks over {len(st)} workgroups; per workgroup total {acc.sum():.0f}")
print(f"   workgroup start offsets: median {np.median(starts):.0f}, 90 % {np.percentile(starts, 90):.0f}, max {starts.max()}")
for n, c in zip(names, acc):
    print(f"   {n:66s} {c:9.0f}  {100 * c / acc.sum():5.1f} %")
