"""MI355X executor for YDB column-shard SSA programs: Python host-side binding over the C-ABI."""
from .arrow_c import Col, HostBatch, ResultColumn  # noqa: F401
from .ssa_pb import AGG, F, YQL, Aggregate, Program  # noqa: F401

__all__ = ["Col", "HostBatch", "ResultColumn", "AGG", "F", "YQL", "Aggregate", "Program", "ProgramContainer", "SsaError"]


def __getattr__(name):
    # The native library is loaded lazily so that pure-host helpers (program builder, Arrow C structs)
    # stay importable on machines without the built extension.
    if name in ("ProgramContainer", "SsaError", "DeviceBatch", "device_batch"):
        from . import program as _program
        return getattr(_program, name)
    raise AttributeError(name)
