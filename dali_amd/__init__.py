"""dali_amd -- MI355X-native (gfx950) implementation of the DALI JPEG -> RandomResizedCrop ->
CropMirrorNormalize hot path behind DALI's Pipeline / fn.* / DALIGenericIterator surface."""
__version__ = "0.1.0"
