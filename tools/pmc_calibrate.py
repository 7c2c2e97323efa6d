"""Known-byte-count streams for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this library's access
shape (one dword per lane).  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pulselib_amd import _native  # noqa: E402

lib = _native.lib()
dev = torch.device("cuda:0")
n_words = 512 * 1024 * 1024 // 4          # 512 MiB: beyond the 256 MiB Infinity Cache
buf = torch.zeros(n_words, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for rep in range(5):
    _native.check(lib.pulse_calib_stream(buf.data_ptr(), n_words, 0, stream))
for rep in range(5):
    _native.check(lib.pulse_calib_stream(buf.data_ptr(), n_words, 1, stream))
torch.cuda.synchronize()
print("bytes per launch", n_words * 4)
