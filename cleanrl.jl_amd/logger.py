"""Logger.make_logger (logger.jl:7-29): tee of console / JSON-lines sinks installed on the "CleanRL" logger.
Records keep the reference's names and keys ("Episode Statistics", "Training Statistics"; ppo.jl:157,247).
TensorBoard output stays a host-language concern (TensorBoardLogger.jl upstream); here it maps to the JSON sink."""
import json
import logging
import os


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
    if to_tensorboard or to_json:
        lg.addHandler(_JsonLines(os.path.join(log_dir, f"{run_name}.json")))
    if not lg.handlers:
        lg.addHandler(logging.NullHandler())
    return lg
