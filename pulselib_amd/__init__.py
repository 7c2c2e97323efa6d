"""pulselib_amd -- MI355X-native batched env-step engine behind Pulselib's environment API.

Hot path only: the Poker GPU step / reset / scripted-opponent policy, and the
Blackjack, 2048 and Particle2D step kernels, as hand-written gfx950 HIP behind the C ABI in
include/pulse_env.h.  PyTorch-ROCm supplies device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
