"""ConfigParser.argparse_struct (utils/config_parser.jl:18-40): every field of a config struct becomes a `--field` command
line option whose default is the struct's value and whose type is the field's type; the parsed values come back in the same
struct type. Like the reference there is no help text per option."""
import argparse
import dataclasses


def _bool(s):
    if isinstance(s, bool):
        return s
    if s.lower() in ("true", "1", "yes"):
        return True
    if s.lower() in ("false", "0", "no"):
        return False
    raise argparse.ArgumentTypeError(f"not a Bool: {s!r}")


def argparse_struct(s, argv=None):
    """config_parser.jl:18-40. `s` is a dataclass instance (the mirror of a Base.@kwdef struct)."""
    if not dataclasses.is_dataclass(s):
        raise TypeError("argparse_struct: expected a config struct (dataclass instance)")
    parser = argparse.ArgumentParser()
    for f in dataclasses.fields(s):
        value = getattr(s, f.name)
        typ = _bool if isinstance(value, bool) else type(value)
        parser.add_argument(f"--{f.name}", default=value, type=typ)
    args = parser.parse_args(argv)
    return type(s)(**vars(args))
