"""capf — ctypes binding of libcapf.so (include/capf.h), the MI355X-native Context-Aware PoseFormer
hot path, plus deterministic synthetic data (capf.synth) and data-parallel helpers (capf.dist).

There is deliberately NO fallback: importing `capf.lib` raises if libcapf.so has not been built
(python __graft_entry__.py, or `make -C contextaware-poseformer_amd/csrc`)."""
from .lib import CapfConfig, Engine, CapfError, load_library, LIB_PATH, HRNET, CPN50  # noqa: F401
