"""bwa-mem-sw on MI355X: HIP (gfx950) implementation of BWA-MEM's banded Smith-Waterman
seed-extension path behind the reference's task/result-batch operator interface.

The directory name contains '-', so import it through `__graft_entry__.load_package()`
(registers this package as `bwa_mem_sw_amd`).  All compute goes through
libbwasw_mi355.so (csrc/); there is no Python or CPU compute path here.
"""
from . import host  # noqa: F401
from .host import (BswContext, BswError, PARAMS, TASK, RESULT, EXT, EXT_TASK, SYNTH,  # noqa: F401
                   default_params, synth_tasks, lib, lib_path, build_library)
