"""``ArrayError`` of the reference (src/lib.rs:9-16) plus the ABI's own failures."""
from __future__ import annotations

from . import _lib


class ArrayError(ValueError):
    """The error type for input arrays (src/lib.rs:9-16)."""


class Empty(ArrayError):
    def __init__(self):
        super().__init__("array is empty")


class NotContiguous(ArrayError):
    def __init__(self):
        super().__init__("array is not contiguous in memory")


ArrayError.Empty = Empty
ArrayError.NotContiguous = NotContiguous


class PetalError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__(f"[{code}] {msg}")


def check(rc: int):
    if rc == _lib.PN_OK:
        return
    if rc == _lib.PN_ERR_EMPTY:
        raise Empty()
    if rc == _lib.PN_ERR_NOT_CONTIGUOUS:
        raise NotContiguous()
    msg = _lib.last_error()
    if rc == _lib.PN_ERR_EMPTY_MATRIX:
        raise AssertionError(msg or "empty matrix")  # the reference panics (src/ball_tree.rs:582)
    raise PetalError(rc, msg)
