"""savont_amd: ctypes harness of the MI355X-native `savont asv` hot path (libsavont_hip.so, libsavont_asv.so)."""
import os

# Several samples in flight are several HIP streams; the runtime maps them onto GPU_MAX_HW_QUEUES hardware queues (default 4), where a long kernel blocks
# every stream that shares its queue.  Must be set before the HIP runtime initialises (torch or the library, whichever comes first); the library sets
# the same default in svt_create for callers that do not come through this package.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
