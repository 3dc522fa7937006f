"""Functional operator API, generated from the operator schemas registered in the C++ host library
(reference: dali/python/nvidia/dali/fn/__init__.py:31-148 -- `a__b__OpName` -> fn.a.b.op_name)."""
import re
import sys
import types as _pytypes

from .. import _backend as _b
from ..data_node import DataNode as _DataNode

_special_case_mapping = {"b_box": "bbox", "mx_net": "mxnet", "tf_record": "tfrecord"}


_WORD_BREAKS = (
    (re.compile(r"([A-Z]+)([A-Z][a-z])"), r"\1_\2"),     # an acronym ends where a capitalised word starts: TFRecord
    (re.compile(r"([a-z])([A-Z])"), r"\1_\2"),            # a word ends at the next capital: CoinFlip
    (re.compile(r"([0-9])([A-Z][a-z])"), r"\1_\2"),       # ... also behind a digit (Caffe2Reader), but Warp3D stays whole
)


def _to_snake_case(pascal):
    """`RandomResizedCrop` -> `random_resized_crop`, the naming rule of the reference's fn module."""
    for pattern, repl in _WORD_BREAKS:
        pascal = pattern.sub(repl, pascal)
    snake = pascal.lower()
    for artifact, desired in _special_case_mapping.items():
        snake = snake.replace(artifact, desired)
    return snake


def _choose_device(inputs):
    for i in inputs:
        if isinstance(i, _DataNode) and i.device == "gpu":
            return "gpu"
        if isinstance(i, (list, tuple)) and any(isinstance(j, _DataNode) and j.device == "gpu" for j in i):
            return "gpu"
    return "cpu"


def _count_of(arg):
    def count(init_args):
        v = init_args.get(arg, [])
        return 1 if isinstance(v, str) else len(v)
    return count


# operators whose number of outputs follows an argument (OpSchema::OutputFn in the reference)
_OUTPUT_COUNT = {"readers__Webdataset": _count_of("ext"), "readers__TFRecord": _count_of("feature_names"),
                 "TFRecordReader": _count_of("feature_names")}


def _make_fn(schema_name, wrapper_name):
    schema = _b.get_schema(schema_name)
    arg_defs = {a["name"]: a for a in schema["args"]}

    def fn_wrapper(*inputs, **kwargs):
        from ..pipeline import Pipeline
        pipe = Pipeline.current()
        if pipe is None:
            raise RuntimeError(f"fn.{wrapper_name} must be called inside a pipeline definition "
                               "(`with pipeline:` or a @pipeline_def function)")
        name = kwargs.pop("name", None)
        kwargs.pop("preserve", None)
        kwargs.pop("bytes_per_sample_hint", None)
        flat_inputs = []
        for i in inputs:
            if isinstance(i, (list, tuple)):
                flat_inputs.extend(i)
            else:
                flat_inputs.append(i)
        for i in flat_inputs:
            if not isinstance(i, _DataNode):
                raise TypeError(f"Inputs of fn.{wrapper_name} must be DataNodes, got {type(i).__name__}. Constant inputs "
                                "are not supported in this build.")
        device = kwargs.pop("device", None) or _choose_device(flat_inputs)
        if device not in ("cpu", "gpu", "mixed"):
            raise ValueError(f'Invalid device "{device}". Valid options are "cpu", "gpu" or "mixed"')
        if device not in schema["backends"]:
            avail = ", ".join(f'"{b}"' for b in schema["backends"]) or "none"
            raise RuntimeError(f'Operator fn.{wrapper_name} is not available for device "{device}" in this '
                               f"MI355X-native build (registered backends: {avail}). There is no CPU fallback for "
                               "device operators.")
        if not (schema["min_inputs"] <= len(flat_inputs) <= schema["max_inputs"]):
            raise ValueError(f"Operator fn.{wrapper_name} expects between {schema['min_inputs']} and "
                             f"{schema['max_inputs']} inputs, but received {len(flat_inputs)}")
        if device in ("cpu", "mixed"):
            for i in flat_inputs:
                if i.device != "cpu":
                    raise ValueError(f'{"CPU" if device == "cpu" else "Mixed"} operator fn.{wrapper_name} cannot take '
                                     "a GPU input")
        else:
            flat_inputs = [i.gpu() for i in flat_inputs]
        init_args, arg_inputs = {}, {}
        for k, v in kwargs.items():
            if v is None:
                continue
            if k not in arg_defs:
                raise TypeError(f"Operator fn.{wrapper_name} got an unexpected keyword argument '{k}'")
            if isinstance(v, _DataNode):
                if not arg_defs[k]["tensor_ok"]:
                    raise TypeError(f"Argument `{k}` of fn.{wrapper_name} cannot be a DataNode (tensor argument)")
                if v.device != "cpu":
                    raise ValueError(f"Tensor argument `{k}` of fn.{wrapper_name} must be a CPU DataNode")
                arg_inputs[k] = v
            else:
                init_args[k] = v
        nout = schema["num_outputs"]
        if schema_name in _OUTPUT_COUNT:      # the reference's OutputFn: the output count follows an argument
            nout = _OUTPUT_COUNT[schema_name](init_args)
        outs = pipe._add_op(schema_name, device, init_args, flat_inputs, arg_inputs, nout, name)
        return outs[0] if len(outs) == 1 else outs     # single outputs are unwrapped (ops/__init__.py:510-514)

    fn_wrapper.__name__ = fn_wrapper.__qualname__ = wrapper_name
    lines = [schema["doc"], "", "Keyword args", "------------"]
    for a in schema["args"]:
        default = "required" if a["required"] else f"optional, default = {a['default']!r}"
        lines.append(f"`{a['name']}` : {a['type']}{' or TensorList' if a['tensor_ok'] else ''}, {default}\n    {a['doc']}")
    fn_wrapper.__doc__ = "\n".join(lines)
    fn_wrapper._schema_name = schema_name
    return fn_wrapper


def _populate():
    root = sys.modules[__name__]
    for schema_name in _b.schema_names():
        info = _b.get_schema(schema_name)
        if info["internal"] or not info["backends"]:
            continue
        *path, op = schema_name.split("__")
        mod = root
        for p in path:
            full = mod.__name__ + "." + p
            if not hasattr(mod, p):
                sub = _pytypes.ModuleType(full)
                setattr(mod, p, sub)
                sys.modules[full] = sub
            mod = getattr(mod, p)
        name = _to_snake_case(op)
        if not hasattr(mod, name):
            setattr(mod, name, _make_fn(schema_name, name))


def external_source(source=None, num_outputs=None, *, cycle=None, name=None, device="cpu", layout=None, dtype=None,
                    ndim=None, batch=True, **kwargs):
    """fn.external_source: data fed from Python.  `source` may be a callable (called once per iteration,
    returns a batch = list/array of samples) or an iterable of batches; without a source the data is provided
    with `Pipeline.feed_input(name, batch)`."""
    from ..pipeline import Pipeline
    pipe = Pipeline.current()
    if pipe is None:
        raise RuntimeError("fn.external_source must be called inside a pipeline definition")
    if num_outputs is not None:
        raise NotImplementedError("external_source with num_outputs is not supported in this build")
    init = {}
    if layout:
        init["layout"] = layout
    if dtype is not None:
        init["dtype"] = int(dtype)
    if ndim is not None:
        init["ndim"] = int(ndim)
    node = pipe._add_op("ExternalSource", "cpu", init, [], {}, 1, name)[0]
    if source is not None:
        state = {"it": iter(source) if not callable(source) else None, "i": 0}

        def feed(p):
            if state["it"] is not None:
                try:
                    batch = next(state["it"])
                except StopIteration:
                    if cycle in (True, "quiet"):
                        state["it"] = iter(source)
                        batch = next(state["it"])
                    else:
                        raise
            else:
                try:
                    nparams = len(__import__("inspect").signature(source).parameters)
                except (TypeError, ValueError):
                    nparams = 0
                batch = source(state["i"]) if nparams >= 1 else source()
                state["i"] += 1
            p.feed_input(node, batch, layout)

        pipe._input_callbacks.append(feed)
    return node.gpu() if device == "gpu" else node


_populate()


def _wrap_tfrecord():
    """fn.readers.tfrecord(path, index_path, features={name: tfrecord.FixedLenFeature / VarLenFeature}) -> dict of
    outputs, like the reference (dali/python/nvidia/dali/ops/_operators/tfrecord.py): the dictionary is flattened
    into the per-feature argument vectors of the operator."""
    root = sys.modules[__name__]
    raw = root.readers.tfrecord

    def tfrecord(*, path, index_path, features, **kwargs):
        names = list(features)
        feats = [features[k] for k in names]
        shapes = [e for f in feats for e in f.shape]
        outs = raw(path=[path] if isinstance(path, str) else list(path),
                   index_path=[index_path] if isinstance(index_path, str) else list(index_path), feature_names=names,
                   feature_dtypes=[int(f.dtype) for f in feats], feature_has_shape=[int(f.has_shape) for f in feats],
                   feature_ndims=[len(f.shape) for f in feats], **({"feature_shapes": shapes} if shapes else {}), **kwargs)
        return dict(zip(names, outs if isinstance(outs, (list, tuple)) else [outs]))

    tfrecord.__doc__ = raw.__doc__
    tfrecord._schema_name = raw._schema_name
    root.readers.tfrecord = tfrecord


_wrap_tfrecord()
