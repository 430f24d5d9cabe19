"""dyk: host-side runtime of the MI355X-native Double-YOLO-Kaist hot path (ctypes over libdyk_hip.so)."""
from . import lib  # noqa: F401
