"""Enums and constants mirroring nvidia.dali.types (dali/python/nvidia/dali/types.py; values from
include/dali/core/dali_data_type.h:45-70 and include/dali/core/common.h:144-175)."""
import numpy as np


class _DALIEnum(int):
    _names = {}

    def __new__(cls, value, name=None):
        obj = int.__new__(cls, value)
        obj._name = name
        return obj

    def __repr__(self):
        return f"<{type(self).__name__}.{self._name}: {int(self)}>"

    __str__ = __repr__

    @property
    def name(self):
        return self._name

    @property
    def value(self):
        return int(self)


class DALIDataType(_DALIEnum):
    pass


class DALIInterpType(_DALIEnum):
    pass


class DALIImageType(_DALIEnum):
    pass


def _mk(cls, **kw):
    for k, v in kw.items():
        e = cls(v, k)
        setattr(cls, k, e)
        globals()[k] = e


_mk(DALIDataType, NO_TYPE=-1, UINT8=0, UINT16=1, UINT32=2, UINT64=3, INT8=4, INT16=5, INT32=6, INT64=7, FLOAT16=8,
    FLOAT=9, FLOAT64=10, BOOL=11, STRING=12)
_mk(DALIInterpType, INTERP_NN=0, INTERP_LINEAR=1, INTERP_CUBIC=2, INTERP_LANCZOS3=3, INTERP_TRIANGULAR=4,
    INTERP_GAUSSIAN=5)
_mk(DALIImageType, RGB=0, BGR=1, GRAY=2, YCbCr=3, ANY_DATA=4)

_NP = {0: np.uint8, 1: np.uint16, 2: np.uint32, 3: np.uint64, 4: np.int8, 5: np.int16, 6: np.int32, 7: np.int64,
       8: np.float16, 9: np.float32, 10: np.float64, 11: np.bool_}


def to_numpy_type(dali_type):
    return _NP[int(dali_type)]


def from_numpy_type(dtype):
    dtype = np.dtype(dtype)
    for k, v in _NP.items():
        if np.dtype(v) == dtype:
            return DALIDataType(k, [n for n, e in vars(DALIDataType).items() if isinstance(e, DALIDataType) and int(e) == k][0])
    raise TypeError(f"Unsupported numpy dtype {dtype}")


class PipelineAPIType:
    BASIC = 0
    ITERATOR = 1
    SCHEDULED = 2
