"""DataNode: a symbolic handle to an operator output inside a pipeline definition
(reference: dali/python/nvidia/dali/data_node.py)."""


class DataNode:
    def __init__(self, name, device="cpu", source=None):
        self.name = name
        self.device = device
        self.source = source

    def __str__(self):
        return f'DataNode(name="{self.name}", device="{self.device}")'

    __repr__ = __str__

    def gpu(self):
        """Transfers the data to the GPU (inserts the CPU->GPU copy node)."""
        if self.device == "gpu":
            return self
        from .pipeline import Pipeline
        pipe = Pipeline.current()
        if pipe is None:
            raise RuntimeError("DataNode.gpu() must be called inside a pipeline definition")
        return pipe._to_gpu(self)

    def cpu(self):
        if self.device == "cpu":
            return self
        raise RuntimeError("GPU->CPU transfers inside the graph are not supported; call `.as_cpu()` on the outputs "
                           "returned by Pipeline.run() instead.")

    def _arith(self, *_):
        raise NotImplementedError("Arithmetic expressions on DataNodes are outside the scope of this build "
                                  "(JPEG -> RandomResizedCrop -> CropMirrorNormalize hot path)")

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = _arith
