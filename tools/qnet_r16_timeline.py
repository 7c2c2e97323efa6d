"""Where the sixteen-rows-per-wavefront action selection (csrc/qnet_rows16.h: qnet_act_r16_kernel) spends its time: the diagnostic
build (`make -C pulselib_amd/csrc stamps`) stores the clock of lane 0 of EVERY wavefront at the phase boundaries; prints, per
workgroup, the segments of wavefront 0 and when the last wavefront finished.  Read the SHARES (the stamps cost time).
    python tools/qnet_r16_timeline.py [n_rows] [learner fraction]"""
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
CUS = torch.cuda.get_device_properties(0).multi_processor_count
WIN = 1024 if N >= 1024 * CUS else 256
n_blocks = min((N + WIN - 1) // WIN, CUS)
buf = torch.zeros((n_blocks, 16, 16), dtype=torch.int64, device=dev)
net = q._net_struct(q.network)


def act(step):
    _native.check(lib.pulse_qnet_act(C.byref(net), s.data_ptr(), 40, N, seat.data_ptr(), 1, C.c_float(0.1), 1, step, 0, acts.data_ptr(), None, None,
                                     mask.data_ptr(), torch.cuda.current_stream().cuda_stream), "act")


for rep in range(3):
    act(rep)
torch.cuda.synchronize()
names = ["0-1 weights into LDS + candidate loads + ballots", "1-2 first barrier", "2-3 list + second barrier", "3-4 row ids, gather issued",
         "4-5 the five layers (incl. the gather's latency)", "5-6 argmax, draw, store", "6-7 to the end of the wavefront's work"]
acc = np.zeros(7)
reps = 5
for rep in range(reps):
    buf.zero_()
    lib.pulse_debug_set_qnet_stamp_buffer(buf.data_ptr())
    act(10 + rep)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    w0 = st[:, 0, :8]
    w0 = w0[(w0[:, 4] > 0) & (w0[:, 7] > 0)]                  # wavefront 0 of workgroups that ran a tile
    acc += np.diff(w0, axis=1).mean(axis=0)
    t0 = st[:, :, 0][st[:, :, 0] > 0].min()
    ends = st[:, :, 7].max(axis=1) - t0                        # when a workgroup's last wavefront finished, from the launch's first stamp
    starts = st[:, 0, 0] - t0
    tiles = (st[:, :, 5] > 0).sum(axis=1)
lib.pulse_debug_set_qnet_stamp_buffer(None)
acc /= reps
print(f"N={N} learner fraction {frac:.3f}: {n_blocks} workgroups, wavefronts with a tile per workgroup: mean {tiles.mean():.1f} max {tiles.max()}")
print(f"   workgroup first stamp after the launch's first: median {np.median(starts):.0f}, max {starts.max()} ticks")
print(f"   workgroup's last wavefront done: median {np.median(ends):.0f}, 90 % {np.percentile(ends, 90):.0f}, max {ends.max()} ticks")
for n, c in zip(names, acc):
    print(f"   {n:58s} {c:9.0f}  {100 * c / acc.sum():5.1f} %")
lastw = st[:, :, 5].max(axis=1) - st[:, 0, 3]
print(f"   list ready -> the workgroup's last tile through the layers: median {np.median(lastw):.0f}, max {lastw.max()} ticks")
