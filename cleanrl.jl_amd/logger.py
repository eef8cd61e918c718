"""Logger.make_logger (logger.jl:7-29): tee of console / TensorBoard / JSON-lines sinks installed on the "CleanRL" logger.
Records keep the reference's names and keys ("Episode Statistics", "Training Statistics"; ppo.jl:157,247).
The TensorBoard sink (logger.jl:14-16: TBLogger("logs/<run_name>")) writes the event-file format itself — TFRecord framing with
masked CRC-32C, Event / Summary protobuf messages encoded by hand — so no TensorFlow or tensorboard package is needed to write,
only to view. Like TensorBoardLogger.jl it logs every numeric key of a record as the scalar "<message>/<key>" and treats
`log_step_increment` as the amount to advance its step by (ppo.jl:155-157), not as a value."""
import json
import logging
import os
import socket
import struct
import time


def _crc32c_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc32c_table()


def _masked_crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    c ^= 0xFFFFFFFF
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int = 0, file_version: str = None, scalars=None) -> bytes:
    """tensorflow.Event: 1 wall_time (double), 2 step (int64), 3 file_version (string), 5 summary {1 value {1 tag, 2 simple_value}}."""
    ev = b"\x09" + struct.pack("<d", wall_time) + b"\x10" + _varint(step)
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b"".join(_field_bytes(1, _field_bytes(1, tag.encode()) + b"\x15" + struct.pack("<f", float(v))) for tag, v in scalars)
        ev += _field_bytes(5, summary)
    return ev


class _TensorBoard(logging.Handler):
    """TBLogger("logs/<run_name>") of logger.jl:15: one events.out.tfevents.* file in that directory."""

    def __init__(self, log_dir):
        super().__init__()
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}")
        self.f = open(self.path, "ab")
        self.step = 0
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, data: bytes):
        head = struct.pack("<Q", len(data))
        self.f.write(head + struct.pack("<I", _masked_crc32c(head)) + data + struct.pack("<I", _masked_crc32c(data)))
        self.f.flush()

    def emit(self, record):
        kv = dict(getattr(record, "crl", {}))
        self.step += int(kv.pop("log_step_increment", 1))
        scalars = [(f"{record.getMessage()}/{k}", v) for k, v in kv.items() if isinstance(v, (int, float)) and not isinstance(v, bool)]
        if scalars:
            self._record(_event(time.time(), self.step, scalars=scalars))

    def close(self):
        try:
            self.f.close()
        finally:
            super().close()


class _JsonLines(logging.Handler):
    def __init__(self, path):
        super().__init__()
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        self.f = open(path, "a")

    def emit(self, record):
        self.f.write(json.dumps({"msg": record.getMessage(), **getattr(record, "crl", {})}) + "\n")
        self.f.flush()


class _Console(logging.StreamHandler):
    def format(self, record):
        kv = " ".join(f"{k}={v}" for k, v in getattr(record, "crl", {}).items())
        return f"[{record.getMessage()}] {kv}"


def make_logger(run_name, to_terminal=True, to_tensorboard=True, to_json=False, log_dir="logs"):
    lg = logging.getLogger("CleanRL")
    for hd in list(lg.handlers):
        lg.removeHandler(hd)
    lg.setLevel(logging.INFO)
    lg.propagate = False
    if to_terminal:
        lg.addHandler(_Console())
    if to_tensorboard:
        lg.addHandler(_TensorBoard(os.path.join(log_dir, run_name)))
    if to_json:
        lg.addHandler(_JsonLines(os.path.join(log_dir, f"{run_name}.json")))
    if not lg.handlers:
        lg.addHandler(logging.NullHandler())
    return lg
