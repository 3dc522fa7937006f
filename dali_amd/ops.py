"""Class-style operator API (`ops.readers.File(...)(...)`), a thin veneer over `fn`
(reference: dali/python/nvidia/dali/ops/__init__.py:553-760, python_op_factory)."""
import sys
import types as _pytypes

from . import _backend as _b
from . import fn as _fn


def _make_op_class(schema_name, fn_callable, class_name):
    class _Op:
        def __init__(self, **init_args):
            self._init_args = init_args

        def __call__(self, *inputs, **call_args):
            return fn_callable(*inputs, **{**self._init_args, **call_args})

    _Op.__name__ = _Op.__qualname__ = class_name
    _Op.__doc__ = fn_callable.__doc__
    _Op.schema_name = schema_name
    return _Op


def _populate():
    root = sys.modules[__name__]
    for schema_name in _b.schema_names():
        info = _b.get_schema(schema_name)
        if info["internal"] or not info["backends"]:
            continue
        *path, op = schema_name.split("__")
        mod, fmod = root, _fn
        for p in path:
            full = mod.__name__ + "." + p
            if not hasattr(mod, p):
                sub = _pytypes.ModuleType(full)
                setattr(mod, p, sub)
                sys.modules[full] = sub
            mod = getattr(mod, p)
            fmod = getattr(fmod, p)
        setattr(mod, op, _make_op_class(schema_name, getattr(fmod, _fn._to_snake_case(op)), op))


_populate()
