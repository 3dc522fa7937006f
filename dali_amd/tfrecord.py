"""Feature descriptions for fn.readers.tfrecord (reference: dali/python/nvidia/dali/tfrecord.py,
dali/operators/reader/parser/tf_feature.h): equivalents of tf.io.FixedLenFeature / tf.io.VarLenFeature."""
from . import types as _types

int64 = _types.INT64
float32 = _types.FLOAT
string = _types.UINT8     # a bytes feature comes out as a 1-D uint8 tensor


class Feature:
    def __init__(self, has_shape, shape, dtype, default_value=None):
        if int(dtype) not in (int(int64), int(float32), int(string)):
            raise TypeError("TFRecord features are tfrecord.int64, tfrecord.float32 or tfrecord.string")
        self.has_shape, self.shape, self.dtype, self.default_value = bool(has_shape), [int(s) for s in shape], dtype, default_value


def FixedLenFeature(shape, dtype, default_value):
    """A feature with a fixed shape ([] = a scalar; strings: [] or [1])."""
    return Feature(True, shape, dtype, default_value)


def VarLenFeature(*args):
    """VarLenFeature(dtype, default_value) or VarLenFeature(partial_shape, dtype, default_value): a feature whose
    (outermost) extent follows the data."""
    if len(args) == 2:
        return Feature(False, [], args[0], args[1])
    if len(args) == 3:
        return Feature(False, args[0], args[1], args[2])
    raise TypeError("VarLenFeature(dtype, default_value) or VarLenFeature(partial_shape, dtype, default_value)")
