"""MI355X-native evaluator for EPA-ng's placement hot path: ctypes surface over the C-ABI of
include/epa_dev.h (api), the C++ host pipeline (hostlib), query sharding (parallel) and the
synthetic workloads of BASELINE.json (synth).  The compute path is libepa_dev.so only."""
from .api import *  # noqa: F401,F403
