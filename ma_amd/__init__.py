"""ma_amd -- MI355X-native seed-and-extend engine behind MA's Module/Pledge API.

Python side = thin ctypes binding of the C ABI in include/ma_amd.h (libma_amd.so, built from
ma_amd/csrc by __graft_entry__.build()).  There is no CPU fallback: every compute call needs a HIP
device and raises MaError otherwise.
"""
from .api import (MaError, Params, Index, Batch, HostArray, lib, lib_path, device_count, set_device, bind_host_thread,
                  ksw_batch, SEGMENT_DT, SEED_DT, EZ_DT, ALIGNMENT_DT, KSW_JOB_DT)

__all__ = ["MaError", "Params", "Index", "Batch", "HostArray", "lib", "lib_path", "device_count", "set_device", "bind_host_thread",
           "ksw_batch", "SEGMENT_DT", "SEED_DT", "EZ_DT", "ALIGNMENT_DT", "KSW_JOB_DT"]
