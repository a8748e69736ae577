"""Module API of the reference's ``models`` package, backed by libdyt_hip.so (MI355X)."""
