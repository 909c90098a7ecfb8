"""unimp_amd -- MI355X-native drop-in for the open_flamingo surface UniMP's mmrec.py uses.

    from unimp_amd import create_model_and_transforms, Flamingo

The arithmetic runs in hand-written HIP kernels (unimp_amd/csrc, C ABI in include/unimp_hip.h)
loaded from the in-tree libunimp_hip.so; there is no eager-PyTorch or CPU compute fallback.
"""
__all__ = ["create_model_and_transforms", "Flamingo"]


def __getattr__(name):
    if name in ("create_model_and_transforms", "Flamingo"):
        from . import factory, flamingo
        return {"create_model_and_transforms": factory.create_model_and_transforms, "Flamingo": flamingo.Flamingo}[name]
    raise AttributeError(name)
