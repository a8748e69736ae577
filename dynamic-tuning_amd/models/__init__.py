"""Module API of the reference's ``models`` package, backed by libdyt_hip.so (MI355X).

The reference's ``models`` is a regular package too, so with this directory in front of the reference root on ``sys.path`` the
modules implemented here (vision_transformer_IN21K, dynamic_adapter, losses) win and the reference's other modules stay
importable under the same package name."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
