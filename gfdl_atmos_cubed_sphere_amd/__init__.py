"""MI355X-native FV3 acoustic-substep hot path: Python host over the C ABI of csrc/libfv3_mi355x.so."""
import os as _os

# HIP maps streams onto 4 hardware queues by default.  The launch stream, the side stream of the sponge-level kernels and
# RCCL's stream then alias, and kernels meant to overlap queue up behind each other (halo exchange next to the interior
# of d_sw: 2.55 ms per step with 4 queues, 2.39 with 8).  Only effective if set before the HIP runtime initialises,
# i.e. before the first torch.cuda / library call of the process; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
